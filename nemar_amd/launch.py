"""One command -> N ranks.  The reference goes multi-GPU from its own flag (`--gpu_ids 0,1,2`, options/base_options.py:127-135 ->
nn.DataParallel, models/networks.py:108-111); this build is one process per GPU over RCCL, so the entry points (bench.py,
nemar_amd.train) re-execute themselves once per rank when they are started WITHOUT a torch.distributed environment and asked for
more than one GPU.  Under torchrun / torch.distributed.run (RANK / WORLD_SIZE already set) nothing is spawned."""
import os
import socket
import subprocess
import sys


def under_launcher():
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_local_ranks(nranks, argv=None, module=None, extra_env=None):
    """Run `python [-m module | script] argv...` once per rank on this node (RANK = LOCAL_RANK = 0..nranks-1, rendezvous on
    127.0.0.1) and wait for all of them.  Returns the first non-zero exit code (0 when every rank succeeded); a failing rank
    takes the others down.  Rank 0 inherits stdout; the other ranks' stdout goes to stderr (a bench prints ONE line)."""
    argv = list(sys.argv[1:] if argv is None else argv)
    cmd = [sys.executable] + (["-m", module] if module else [os.path.abspath(sys.argv[0])]) + argv
    port = str(_free_port())
    procs = []
    for r in range(nranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nranks), "LOCAL_WORLD_SIZE": str(nranks),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "NEMAR_SPAWNED": "1",
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        env.update(extra_env or {})
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:          # one rank died: the collectives of the others would hang
                        q.terminate()
            if pending:
                try:
                    pending[0].wait(timeout=0.2)
                except subprocess.TimeoutExpired:
                    pass
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc
