"""The C-ABI library builds, loads and exports every symbol include/gpmi.h declares.
No compute calls (no GPU needed)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "gpmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gpmi_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    names = _declared()
    for required in ("gpmi_rbf", "gpmi_fit", "gpmi_factorize", "gpmi_predict", "gpmi_get_alpha",
                     "gpmi_post_chol", "gpmi_lml_batch", "gpmi_last_error", "gpmi_get_timers",
                     "gpmi_ctx_create", "gpmi_ctx_destroy"):
        assert required in names


def test_library_exports_every_declared_symbol():
    from gaussian_process_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with `python -m gaussian_process_amd.build`"
    lib = C.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), "libgpmi355x.so lacks %s" % name


def test_ctypes_table_covers_the_header():
    from gaussian_process_amd import _lib
    lib = _lib.load()
    assert lib.gpmi_abi_version() == _lib.ABI_VERSION
    declared = set(_declared()) - {"gpmi_last_error"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_no_silent_cpu_fallback_without_gpu():
    """Without a device the product path must fail loudly, never compute on the CPU."""
    from gaussian_process_amd import _lib
    lib = _lib.load()
    n = C.c_int(-1)
    status = lib.gpmi_device_count(C.byref(n))
    if status == _lib.GPMI_OK and n.value > 0:
        pytest.skip("a GPU is present")
    from gaussian_process_amd import GPContext
    from gaussian_process_amd.GP_regression import RBF_kernel
    with pytest.raises(RuntimeError):
        GPContext(0)
    with pytest.raises(RuntimeError):
        RBF_kernel(np.zeros((4, 2)), np.zeros((3, 2)), 1.0, 1.0)


def test_missing_library_raises(monkeypatch):
    from gaussian_process_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "does_not_exist.so"))
    with pytest.raises(_lib.GpmiLibraryMissing):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gaussian_process_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "gp_oracle" not in text and "rbf_oracle" not in text, f


def test_rccl_binding_resolves_without_linking():
    """gpmi_comm_*: libgpmi355x.so does not link librccl (it must load on hosts without it); the first call opens it at
    run time -- the copy already in the process when PyTorch brought one -- and reports which; creating a communicator
    without a GPU fails with a status, never a crash"""
    import subprocess
    from gaussian_process_amd import _lib
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" not in needed and "libtorch" not in needed
    lib = _lib.load()
    buf, ver = C.create_string_buffer(512), C.c_int()
    assert lib.gpmi_comm_library(buf, 512, C.byref(ver)) == _lib.GPMI_OK, _lib.last_error()
    assert b"rccl" in buf.value and ver.value > 20000
    n = C.c_int(-1)
    if lib.gpmi_device_count(C.byref(n)) == _lib.GPMI_OK and n.value > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    idb = C.create_string_buffer(128)
    assert lib.gpmi_comm_create(idb, 0, 1, 0, C.byref(h)) == _lib.GPMI_ERR_RUNTIME
    assert lib.gpmi_comm_create(idb, 3, 2, 0, C.byref(h)) == _lib.GPMI_ERR_BAD_ARG
