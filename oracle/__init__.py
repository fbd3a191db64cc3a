"""CPU oracle for the DDPM UNet denoiser + DDPM/DDIM sample/train path.

TEST INFRASTRUCTURE ONLY.  This package is a from-scratch, functional (state-dict
in, tensor out) CPU restatement of the reference algorithm for the hot path.  It
may be imported only by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the reported CPU baseline,
never as the thing shipped or measured.  The product (``dmme_amd``) never imports it
and fails loudly when its HIP library is missing.

Parity pin: the oracle is checked against golden vectors produced by importing the
reference itself (``tests/golden/make_golden.py``; fixtures committed under
``tests/golden/``) -- see ``tests/test_oracle_golden.py``.

The arithmetic itself (conv / group-norm / softmax / matmul) lives in the third-party
``torch`` the reference depends on (un-pinned in reference ``setup.py:19-29``); the
oracle calls the same ``torch.nn.functional`` primitives on CPU in fp32.
"""

from .unet import UNetConfig, build_graph, param_table, make_state_dict, unet_forward  # noqa: F401
from .diffusion import (  # noqa: F401
    linear_beta,
    alpha_tables,
    tau_table,
    q_sample,
    ddpm_step,
    ddim_step,
    training_loss,
    ddpm_generate,
    ddim_generate,
)
