"""`-m gpu`: full NEMARModel.optimize_parameters() on the MI355X kernels vs the CPU oracle and vs the golden
fixtures recorded from the reference — losses, warped images, regulariser, gradients and post-Adam weights.
Tolerances (fp32, different summation order than MKLDNN): losses 2e-4 relative, images 2e-3 abs (values in [-1,1]),
per-tensor gradients 5e-3 relative to the tensor's max, <2% of weights taking a different Adam sign step."""
import pytest

from step_configs import STEP_CONFIGS
import step_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(STEP_CONFIGS))
def test_step_parity(name):
    step_parity.run(name, check=True)
