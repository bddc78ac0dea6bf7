"""CPU oracle for the Swin-UNETR hot path (TEST INFRASTRUCTURE ONLY).

This package is a plain PyTorch fp32 restatement, in gather-index form, of the
reference's Swin-UNETR forward path (backward comes from autograd over it).
It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``mivp_amd``) never imports or calls anything in here and fails loudly when
its HIP extension is missing.

Pinning: every function is checked against golden vectors generated in the
build container by importing the reference's own modules (see
``tests/golden/gen_golden.py`` and ``tests/test_oracle_golden.py``).  Results
that depend on MONAI factories (absent from the image) are marked
"parity unpinned at the MONAI boundary" where they are tested.
"""
from .swin_ref import (  # noqa: F401
    rel_pos_bias,
    shift_mask,
    window_attention,
    swin_block,
    swin_pair,
    patch_merge,
    up_block,
)
from .unetr_ref import OracleSwinUnetR, default_conf  # noqa: F401
from .loss_ref import dice_focal_loss, dice_coefficient, mean_iou  # noqa: F401
