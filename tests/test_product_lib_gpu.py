"""`-m gpu`: the PRODUCT library (nemar_amd/lib/libnemar_hip.so — no nemar_tune*, measurement switches compiled in as constants, the
non-default kernels absent) runs the training step, and bit for bit like the measurement build the rest of the test session is bound
to (tests/conftest.py).  Each library gets a fresh process (a process binds one of them for good): tests/product_step.py.

The kernels are the same instruction streams in both builds (tests/test_abi.py compares them); what this adds is the product's host
side — routing with the switches folded to their defaults — on real shapes: the bench workload (batch 8: every conv route, the side inputs, the
weight-pack plans), the multi-resolution discriminators, the reference's default affine / non-square geometry, and 512x512 with the
bilateral multi-resolution regulariser."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def run(name, steps, ab):
    env = dict(os.environ, NEMAR_AB_LIBRARY="1" if ab else "0")
    env.pop("NEMAR_TUNE", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "product_step.py"), name, str(steps)], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("name,steps", [("c2_b8", 3), ("c3_full", 2), ("default_full", 2), ("c4_full", 1)])
def test_product_library_steps_bitwise_like_the_measurement_build(name, steps):
    p, a = run(name, steps, False), run(name, steps, True)
    assert p["library"] == "libnemar_hip.so" and not p["has_switches"]
    assert a["library"] == "libnemar_hip_ab.so" and a["has_switches"]
    assert p["losses"] == a["losses"]
    assert p["buffers"] == a["buffers"]


def test_product_library_refuses_a_switch_on_the_gpu_box():
    env = dict(os.environ, NEMAR_AB_LIBRARY="0")
    env.pop("NEMAR_TUNE", None)
    code = ("import sys; sys.path.insert(0, %r); from nemar_amd import ops, _lib\n"
            "try:\n    ops.tune(20, 0)\nexcept _lib.NemarHipError as e:\n    print('REFUSED', e)\n" % os.path.dirname(HERE))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "REFUSED" in r.stdout and "NEMAR_AB" in r.stdout, (r.stdout, r.stderr[-500:])


SOAK_STEPS = 300


@pytest.mark.parametrize("name", ["c2_b8", "c3_full", "c5_full"])
def test_side_stream_soak_product_library_dropout_on(name):
    """What bench.py times — product library, dropout on, weight-gradient branch on the side stream — against the single-stream order:
    SOAK_STEPS consecutive steps, every step's three gradient buffers compared on the device (tests/side_stream_soak.py; the reference's
    step is repeatable, models/nemar_model.py:266-288).  Round 5 had this only as a tool (tools/diag_step_events.py) and only without dropout."""
    env = dict(os.environ, NEMAR_AB_LIBRARY="0", NEMAR_SIDE_STREAM="1")
    env.pop("NEMAR_TUNE", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "side_stream_soak.py"), name, str(SOAK_STEPS)], env=env, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["library"] == "libnemar_hip.so" and d["dropout"], d
    assert d["single_stream_repeats"] == 5, d
    assert d["steps"] >= SOAK_STEPS and d["events"] == 0, d
