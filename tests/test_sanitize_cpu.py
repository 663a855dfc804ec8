"""CPU-only sanitizer job: what can run without a GPU runs under gcc's AddressSanitizer and
UndefinedBehaviorSanitizer -- the C part of the oracle on edge shapes -- and the C-ABI header is compiled as strict C and
C++ so that nothing in it depends on a HIP toolchain.  No GPU sanitizers exist on this pool (and none are attempted)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

GCC = shutil.which("gcc")


@pytest.mark.skipif(GCC is None, reason="needs gcc")
def test_oracle_c_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "rbf_oracle_san")
    cmd = [GCC, "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-ffp-contract=off", "-fopenmp", "-o", exe, os.path.join(ROOT, "oracle", "rbf_oracle.c"),
           os.path.join(ROOT, "tests", "sanitize", "rbf_oracle_driver.c"), "-lm"]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               OMP_NUM_THREADS="2")
    p = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
    assert "rbf_oracle_driver: ok" in p.stdout


@pytest.mark.skipif(GCC is None, reason="needs gcc")
@pytest.mark.parametrize("lang,std", [("c", "c99"), ("c++", "c++17")])
def test_c_abi_header_is_plain_c_and_cxx(tmp_path, lang, std):
    """include/gpmi.h is the boundary a maintainer binds from any language: it must compile on its own, as C and as C++,
    with every warning on"""
    src = tmp_path / ("t.c" if lang == "c" else "t.cpp")
    src.write_text('#include "gpmi.h"\nint main(void) { return GPMI_OK; }\n')
    subprocess.check_call([GCC if lang == "c" else shutil.which("g++") or GCC, "-x", lang, "-std=" + std, "-Wall", "-Wextra",
                           "-Wpedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o",
                           str(tmp_path / "t.o")])


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_planners_under_asan_ubsan(tmp_path):
    """The product's host-side launch planning (gaussian_process_amd/csrc/gpmi_plan.h: supertile / triangular / staircase
    enumeration with its fixed prefix table, live-tile tests, block-width schedules, the bench line's flop counts) is the
    same header the kernels are compiled from; here plain g++ builds it with AddressSanitizer + UBSan and
    tests/sanitize/plan_check.cpp holds ~16 000 plans against brute force: every live tile enumerated exactly once, no
    dead one, M up to 131072, every supertile edge, ragged sizes, a staircase one row too tall for its table."""
    exe = str(tmp_path / "plan_check")
    subprocess.check_call([shutil.which("g++"), "-O1", "-g", "-std=c++17", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-Wall", "-Wextra", "-Werror", "-I",
                           os.path.join(ROOT, "gaussian_process_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "sanitize", "plan_check.cpp")])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
    assert "plan_check: ok" in p.stdout
