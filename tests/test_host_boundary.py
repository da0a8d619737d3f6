"""The drop-in boundary under OpenCV's ownership rules (CPU, no GPU): host/*.cpp is compiled against cv_shim.h —
whose cv::Mat(rows, cols, type, void*) is a NON-owning header exactly like OpenCV's — together with a caller that
repeats ApplyMultiH's sequence (M/main.cpp:262-296, DrawClusters included) and a do-nothing stand-in for the
engine library, and run under AddressSanitizer.  A Mat header left pointing into a dead buffer (the bug the
round-1 review found at host/MultiH.cpp:163) is a heap-use-after-free here."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "multi-h_amd", "host")
pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ missing")


def _compile(tmp_path, sources, out, extra=()):
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread", "-ffp-contract=off", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-I" + HOST, "-I" + os.path.join(ROOT, "include"),
           *extra, *sources, "-o", str(tmp_path / out)]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600)


def test_apply_multih_sequence_is_memory_clean_under_opencv_ownership_rules(tmp_path):
    srcs = [os.path.join(ROOT, "tests", "apply_multih_caller.cpp"), os.path.join(ROOT, "tests", "fake_engine.cpp"),
            os.path.join(HOST, "MultiH.cpp"), os.path.join(HOST, "merge_step.cpp"), os.path.join(HOST, "approx_neighbours.cpp")]
    b = _compile(tmp_path, srcs, "caller")
    if b.returncode != 0 and ("cannot find" in b.stderr or "unrecognized" in b.stderr):
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-4000:]
    # (the stand-in engine takes its behaviour at creation, so engines are not reused from one MultiH to the next here)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", MULTIH_ENGINE_POOL="0")
    r = subprocess.run([str(tmp_path / "caller")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "apply_multih_caller ok" in r.stdout
    # the four runs: both front-half routes x {two planes survive, one plane -> degenerate tail with original indexing}
    lines = [l for l in r.stdout.splitlines() if l.startswith("mode")]
    assert len(lines) == 4
    assert "labels 60" in lines[2] and "labels 60" in lines[3], "the degenerate tail labels the ORIGINAL points"
    assert "labels 48" in lines[1], "the GPU front half drops the filtered points (fake engine: two of every ten)"


_ALIAS = r'''
#include "cv_shim.h"
int main() { const double v[4] = {1, 2, 3, 4}; cv::Mat m(2, 2, CV_64F, v); return (int)m.at<double>(0, 0); }
'''


def test_a_header_over_const_memory_does_not_compile(tmp_path):
    """cv::Mat(rows, cols, type, ptr) takes a pointer to NON-const: the pattern the host code used in round 1
    (a Mat built straight from a `const double*`) must be rejected at compile time, as real OpenCV rejects it."""
    src = tmp_path / "alias.cpp"
    src.write_text(_ALIAS)
    b = _compile(tmp_path, [str(src)], "alias")
    assert b.returncode != 0 and "invalid conversion" in b.stderr


def test_shim_mat_semantics():
    """Header copies share storage, clone() does not, an external-pointer Mat owns nothing."""
    code = r'''
#include <cassert>
#include "cv_shim.h"
int main() {
    cv::Mat a(2, 2, CV_64F); a.at<double>(1, 1) = 5.0;
    cv::Mat b = a; b.at<double>(1, 1) = 6.0; assert(a.at<double>(1, 1) == 6.0);
    cv::Mat c = a.clone(); c.at<double>(1, 1) = 7.0; assert(a.at<double>(1, 1) == 6.0);
    double ext[4] = {1, 2, 3, 4}; cv::Mat d(2, 2, CV_64F, ext); ext[3] = 9.0; assert(d.at<double>(1, 1) == 9.0);
    cv::Mat img(8, 8, CV_8UC3); cv::circle(img, cv::Point2d(4, 4), 2, cv::Scalar(10, 20, 30), -1);
    assert(img.data[3 * (4 * 8 + 4) + 1] == 20 && img.data[0] == 0);
    return 0; }
'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "sem.cpp")
        open(p, "w").write(code)
        b = subprocess.run(["g++", "-std=c++17", "-I" + HOST, p, "-o", os.path.join(d, "sem")], capture_output=True, text=True)
        assert b.returncode == 0, b.stderr
        assert subprocess.run([os.path.join(d, "sem")]).returncode == 0
