"""CPU tests of the drop-in boundary: the library loads, exports exactly what
include/multih_hip.h declares, refuses to run without a GPU (no CPU fallback), and the
product tree never reaches into oracle/."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "multih_hip.h")).read()
    return re.findall(r"^MH_API\s+[\w\s\*]+?\b(mh_\w+)\s*\(", text, flags=re.M)


def test_header_symbols_match_binding_and_library(mh, engine_lib):
    declared = _declared()
    assert len(declared) == len(set(declared)) >= 30
    assert sorted(declared) == sorted(mh.SYMBOLS), "capi.SYMBOLS out of sync with the header"
    for s in declared:
        assert hasattr(engine_lib, s), f"{s} declared in the header but not exported"
    out = subprocess.run(["nm", "-D", "--defined-only", mh.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (mh_\w+)", out))
    assert exported == set(declared), "library exports symbols the header does not declare (or vice versa)"


def test_abi_version_and_error_string(engine_lib):
    assert engine_lib.mh_abi_version() == 2
    assert isinstance(engine_lib.mh_last_error(), (bytes, type(None)))
    assert engine_lib.mh_device_count() >= 0


def test_no_cpu_fallback(mh, engine_lib):
    if engine_lib.mh_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(mh.MultiHError) as ei:
        mh.Engine()
    assert ei.value.code == -1 and "no CPU fallback" in str(ei.value)
    # null-handle calls fail with a status, they never crash or compute
    assert engine_lib.mh_score(None, ctypes.c_double(1.0), None, None) == -2
    assert engine_lib.mh_data_cost(None, None) == -2


def test_host_library_exports_class_hooks(mh, engine_lib):
    host = os.path.join(os.path.dirname(mh.LIB_PATH), "libmultih_host.so")
    assert os.path.exists(host), "host layer not built"
    lib = ctypes.CDLL(host)
    for s in ("mhh_run_process", "mhh_mean_shift", "mhh_homography_3pt", "mhh_homography_3pt_refined",
              "mhh_compatibility_check", "mhh_compatibility_medians_on_engine", "mhh_homography_features", "mhh_set_sharding", "mhh_set_sharding_stream", "mhh_set_device",
              "mhh_set_neighbourhood", "mhh_set_post_filter", "mhh_set_engine_tuning"):
        assert hasattr(lib, s)
    out = subprocess.run(["nm", "-DC", "--defined-only", host], capture_output=True, text=True).stdout
    for method in ("MultiH::Process(", "MultiH::MultiH(", "MultiH::Release()"):
        assert method in out


def test_rccl_transport_library_exports_its_header(mh, engine_lib):
    """libmultih_rccl.so (the native multi-GPU transport) exports exactly what include/multih_rccl.h declares and links
    RCCL; the engine and the host class do not depend on it (they take the transport as a function pointer)."""
    text = open(os.path.join(ROOT, "include", "multih_rccl.h")).read()
    declared = set(re.findall(r"\b(mhr_\w+)\s*\(", text))
    lib_path = os.path.join(os.path.dirname(mh.LIB_PATH), "libmultih_rccl.so")
    assert os.path.exists(lib_path), "RCCL transport not built"
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (mhr_\w+)", out))
    assert exported == declared and "mhr_allgather" in exported
    needed = subprocess.run(["readelf", "-d", lib_path], capture_output=True, text=True).stdout
    assert "librccl" in needed
    for other in ("libmultih_hip.so", "libmultih_host.so"):
        d = subprocess.run(["readelf", "-d", os.path.join(os.path.dirname(mh.LIB_PATH), other)], capture_output=True, text=True).stdout
        assert "librccl" not in d and "libmultih_rccl" not in d


def test_product_tree_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under multi-h_amd/ or include/ may import,
    link or dlopen it (a product path through the oracle would void every parity claim)."""
    bad = []
    for base in ("multi-h_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if "_build" in dp or "__pycache__" in dp:
                continue
            for f in files:
                if not f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    continue
                text = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"oracle_lib|libmh_oracle|libmh_ref_gco|mho_\w+\(|#include\s+\"[^\"]*oracle", text):
                    bad.append(os.path.join(dp, f))
    assert not bad, f"product files reference the oracle: {bad}"
    out = subprocess.run(["ldd", os.path.join(ROOT, "multi-h_amd", "libmultih_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_product_library_kernels_are_the_product_instantiations_only(mh, engine_lib):
    """the kernel templates carry measurement switches (residual_wg: CALIB = store-only calibration,
    CONTRACT = fused multiply-adds — not bit-exact —, TILED = tile-major R, SF = asm store flavours; k_cost32: BATCH;
    k_score32: tilings) — none of them may be instantiated in the product library.  The kernel names the host side
    registers are strings of the .so: demangle them and look at the template arguments."""
    raw = subprocess.run(["strings", mh.LIB_PATH], capture_output=True, text=True).stdout
    names = sorted({l for l in raw.splitlines() if re.match(r"^_ZN2mh\d+k_", l)})
    assert len(names) > 40
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    kernels = sorted({re.sub(r"\(.*", "", d).replace("void ", "") for d in dem})
    res = [k for k in kernels if k.startswith("mh::k_residual")]
    assert 5 <= len(res) <= 8, res
    for k in res:
        a = [x.strip() for x in k[k.index("<") + 1:k.rindex(">")].split(",")]
        # <PPL, MC, WRITE_R, MASK, NT, FAST, CALIB, HSGPR, SYM, CONTRACT, LEAN, TILED, SF, SEMI, MINW>
        # (MC: 64 models per work item in the materialising sweep since r05, 16 in the store-free and symmetric forms)
        assert len(a) == 15 and a[0] == "4" and a[1] in ("16", "64"), k
        assert a[6] == "false" and a[9] == "false" and a[11] == "false" and a[12] == "0" and a[14] == "1", f"measurement variant in the product library: {k}"
        assert a[5] == "true", f"compiler division in the product library: {k}"
    cost = [k for k in kernels if k.startswith("mh::k_cost32")]
    assert sorted(cost) == ["mh::k_cost32<32, 8, false>", "mh::k_cost32_resident<32, 8, false>"], cost
    score = [k for k in kernels if k.startswith("mh::k_score32")]
    assert score and all(k.endswith("<4, 64, false, 6>") or k.endswith("<4, 64, true, 6>") for k in score), score
    # ... while the measurement library (when it has been built) does carry them
    tuning = os.path.join(os.path.dirname(mh.LIB_PATH), "libmultih_hip_tuning.so")
    if os.path.exists(tuning) and os.path.getmtime(tuning) >= os.path.getmtime(mh.LIB_PATH):
        traw = subprocess.run(["strings", tuning], capture_output=True, text=True).stdout
        assert len({l for l in traw.splitlines() if re.match(r"^_ZN2mh\d+k_residual", l)}) > 20
