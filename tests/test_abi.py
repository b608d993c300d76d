"""CPU tier: the C-ABI library loads and exports every symbol include/t2l.h declares (no compute calls)."""
import os.path as osp
import re

import pytest

REPO = osp.dirname(osp.dirname(osp.abspath(__file__)))


def _declared():
    src = open(osp.join(REPO, "include", "t2l.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(t2l_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_boundary():
    names = _declared()
    for need in ("t2l_create", "t2l_destroy", "t2l_load_weights", "t2l_encode_cells", "t2l_db_set", "t2l_search",
                 "t2l_contrastive_loss", "t2l_last_error"):
        assert need in names


def test_library_exports_every_declared_symbol():
    from text2loc_amd import engine

    lib = engine.load_library()
    for name in _declared():
        assert hasattr(lib, name), f"libt2l.so does not export {name}"
    assert set(engine.EXPORTS) == set(_declared()), "ctypes binding and header disagree"
    assert lib.t2l_abi_version() == 2


def test_null_context_is_rejected_without_a_gpu():
    from text2loc_amd import engine

    lib = engine.load_library()
    assert lib.t2l_search(None, None, 1, 1, None, None, None) == -1  # T2L_EINVAL
    assert lib.t2l_db_rows(None) == -1
    assert lib.t2l_last_error(None) == b"null context"


def test_engine_refuses_to_run_without_gpu():
    import torch
    from text2loc_amd import engine

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.T2LError):
        engine.Engine()
