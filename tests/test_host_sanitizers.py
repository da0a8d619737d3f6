"""AddressSanitizer + UBSan and ThreadSanitizer runs of the host-side C++ (multi-h_amd/host/merge_step.cpp)
through tests/host_sanitize_driver.cpp.  GPU ASan is not available on the MI355X pool, so the CPU build
is where the sanitizers run; the driver covers the feature map, mean shift, the 3-point solver with its
LM refinement, the threaded compatibility check and (r05) the FLANN-like neighbourhood builder (approx_neighbours.cpp)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "multi-h_amd", "host")


def _build_and_run(tmp_path, san, args=()):
    exe = str(tmp_path / f"driver_{san.replace(',', '_')}")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread", "-ffp-contract=off", f"-fsanitize={san}",
           "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-I" + HOST,
           os.path.join(ROOT, "tests", "host_sanitize_driver.cpp"), os.path.join(HOST, "merge_step.cpp"),
           os.path.join(HOST, "approx_neighbours.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and ("cannot find" in b.stderr or "unrecognized" in b.stderr):
        pytest.skip(f"-fsanitize={san} runtime not installed: {b.stderr[-200:]}")
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:] + r.stderr[-3000:])
    assert "checksum" in r.stdout
    return [l for l in r.stdout.splitlines() if l.startswith("checksum")][0]


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ missing")
def test_host_code_is_clean_under_asan_ubsan_and_tsan(tmp_path):
    full = _build_and_run(tmp_path, "address,undefined")
    # ThreadSanitizer on the threaded part; same inputs, so the same medians
    threads = _build_and_run(tmp_path, "thread", args=("threads-only",))
    assert full and threads
