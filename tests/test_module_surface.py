"""CPU: the product's SwinUnetR keeps the reference's Python surface:
state_dict keys / shapes / dtypes / order, named_parameters order, which parameters are frozen per
mode and what each named_parameters_* helper returns -- all compared with lists captured from the
reference itself (tests/golden/unetr_*.npz meta).  Also: the C-ABI library loads and exports every
symbol include/mivp.h declares, and the product refuses CPU tensors instead of falling back."""
import os
import re
from argparse import Namespace

import pytest
import torch

from conftest import load_fixture, ROOT

UNETR = ["downstream_e0d0", "downstream_e0d1", "downstream_e1d0", "downstream_e1d1",
         "self_supervised_learning_all_e1d0", "self_supervised_learning_decoder_e1d1",
         "supervised_learning_all_e0d0", "downstream_e1d1_simple"]


@pytest.mark.parametrize("tag", UNETR)
def test_state_dict_and_groups_match_reference(tag):
    import mivp_amd
    from mivp_amd.swin_unetr import SwinUnetR
    fx = load_fixture(f"unetr_{tag}")
    conf = Namespace(**fx.meta["conf"])
    model = SwinUnetR(conf=conf) if tag.endswith("d1") else SwinUnetR(conf)     # positional and keyword ctor
    sd = model.state_dict()
    got = [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()]
    assert got == fx.meta["state_keys"]
    assert [k for k, _ in model.named_parameters()] == fx.meta["param_order"]
    assert [k for k, q in model.named_parameters() if q.requires_grad] == fx.meta["trainable"]
    ids = {id(q): k for k, q in model.named_parameters()}
    for gname, want in fx.meta["groups"].items():
        if isinstance(want, str):
            with pytest.raises(Exception):
                getattr(model, gname)()
            continue
        assert [ids[id(q)] for _, q in getattr(model, gname)()] == want, gname
    # a reference checkpoint loads strictly
    model.load_state_dict({k: v for k, v in fx["sd"].items()}, strict=True)


def test_bad_mode_and_heads_raise_like_reference():
    import mivp_amd
    from mivp_amd.swin_unetr import SwinUnetR, WindowAttention
    fx = load_fixture("unetr_downstream_e0d0")
    conf = Namespace(**fx.meta["conf"])
    conf.training_mode = "nope"
    with pytest.raises(ValueError):
        SwinUnetR(conf)
    with pytest.raises(ValueError):
        WindowAttention(dim=10, num_heads=4)


def test_cpu_input_is_refused_not_emulated():
    import mivp_amd
    from mivp_amd.swin_unetr import SwinUnetR
    fx = load_fixture("unetr_downstream_e0d0")
    model = SwinUnetR(Namespace(**fx.meta["conf"]))
    with pytest.raises(RuntimeError, match="no CPU"):
        model(torch.rand(1, 1, 16, 16, 16))


def test_cabi_exports_every_declared_symbol():
    import ctypes
    import mivp_amd
    from mivp_amd import _lib
    text = open(os.path.join(ROOT, "include", "mivp.h")).read()
    names = sorted(set(re.findall(r"\b(mivp_[a-z0-9_]+)\s*\(", text)))
    assert len(names) > 20
    lib = _lib.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.mivp_abi_version() == _lib.ABI_VERSION
    # descriptor layouts agree between the header's compiler and the ctypes mirror
    for which, cls in enumerate([_lib.SwinDesc, _lib.MergeDesc, _lib.ConvDesc, _lib.EmbedDesc, _lib.UpcatDesc,
                                 _lib.OperandDesc, _lib.GemmTnDesc]):
        assert lib.mivp_sizeof_desc(which) == ctypes.sizeof(cls), cls.__name__
