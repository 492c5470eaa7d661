"""CPU suite: the C-ABI library builds for gfx950, loads, and exports every symbol
include/archi_knn.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    from archi_amd import _lib
    return _lib


def test_header_symbols_all_exported(built):
    hdr = open(os.path.join(ROOT, "include", "archi_knn.h")).read()
    declared = set(re.findall(r"\b(ak_[a-z0-9_]+)\s*\(", hdr))
    bound = {name for name, _, _ in built.SYMBOLS}
    assert declared == bound, f"header/binding mismatch: {declared ^ bound}"
    lib = built.load()          # binds every symbol; AttributeError if one is missing
    assert lib.ak_version().startswith(b"archi_hip")


def test_product_path_fails_loudly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from archi_amd import HipBackendError
    from archi_amd.index import HipIndex
    with pytest.raises(HipBackendError):
        HipIndex(8, 16)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "archi_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "libknn_oracle" not in src and "oracle/_build" not in src, f


def test_product_library_reads_no_environment_on_call_paths_and_knows_no_wrong_result_switch(built):
    """Round-4 review: getenv() on request paths (undefined behaviour against a concurrent setenv) and WRONG-RESULTS ablation
    switches live in the product library. Now: the AK_* environment is snapshot once at dlopen (csrc/switches.h) -- the product
    library does not even IMPORT getenv --, and the ablation names exist only in libarchi_hip_dbg.so."""
    import subprocess
    lib = os.path.join(ROOT, "archi_amd", "lib", "libarchi_hip.so")
    dbg = os.path.join(ROOT, "archi_amd", "lib", "libarchi_hip_dbg.so")
    undefined = subprocess.run(["nm", "-D", "--undefined-only", lib], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert not re.search(r"\bU (secure_)?getenv\b", undefined), "libarchi_hip.so imports getenv"
    raw = open(lib, "rb").read()
    for name in (b"AK_SCAN_ABLATE", b"AK_TAIL_ABLATE", b"AK_GEMM_ABLATE", b"AK_FFN_ABLATE", b"AK_QKV_DBG", b"AK_ENC_NOFFN"):
        assert name not in raw, name
    assert b"AK_SCAN_ABLATE" in open(dbg, "rb").read()          # ... and do exist where the instrumented kernels are
    assert built.load().ak_debug_set(b"AK_TAIL_ABLATE", b"1") != 0      # the product library refuses the name


@pytest.mark.parametrize("target,binary", [("tsan", "index_host_tsan"), ("asan-index", "index_host_asan")])
def test_index_host_side_under_sanitizers(target, binary):
    """the host-side C++ under sanitizers in a CPU-only target (GPU sanitizers are not available on the
    MI355X pool). `make tsan` / `make asan-index` build tests/native/index_host_tsan_main.cpp -- the request coalescer of
    ak_index_search (csrc/coalesce.h: leader promotion, requests on other threads' stacks) and the layout-epoch protocol of
    the filtered search -- with -fsanitize=thread and -fsanitize=address,undefined; 32 searcher threads, 1 writer, the index
    state destroyed while idle. A sanitizer report or a wrong answer makes the binary exit non-zero."""
    import subprocess
    csrc = os.path.join(ROOT, "archi_amd", "csrc")
    subprocess.check_call(["make", "-s", "-C", csrc, target])
    p = subprocess.run([os.path.join(csrc, "build", binary)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode("utf-8", "replace")
    assert p.returncode == 0 and out.startswith("ok:") and "Sanitizer" not in out, out[-3000:]


@pytest.mark.parametrize("target,binary", [("asan-book", "index_book_asan"), ("tsan-book", "index_book_tsan")])
def test_index_bookkeeping_under_sanitizers(target, binary):
    """The rest of the index's host side -- id <-> slot map (lazy for generated rows), tombstones, the fits / reclaim / grow plan,
    next_id and the layout epoch: csrc/index_book.h, the base of index.hip's Index -- driven alone by
    tests/native/index_book_main.cpp: random adds / generated blocks / removes / re-adds / compactions against a dictionary
    model with host arrays standing in for the device side (the epoch must move exactly when the slot numbering does, a mask
    bound to the old (slots, epoch) must be refused), then 8 readers under the shared lock against one writer the way
    Index::mu is used. -fsanitize=address,undefined and -fsanitize=thread; any report or failed check exits non-zero."""
    import subprocess
    csrc = os.path.join(ROOT, "archi_amd", "csrc")
    subprocess.check_call(["make", "-s", "-C", csrc, target])
    p = subprocess.run([os.path.join(csrc, "build", binary)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode("utf-8", "replace")
    assert p.returncode == 0 and out.rstrip().endswith("index_book: ok") and "Sanitizer" not in out, out[-3000:]


def test_scan_filter_reads_mfma_results_only_behind_the_settle_fence(tmp_path):
    """Round-5 advisor finding: the main-pass filter reads raw MFMA accumulators through inline-asm v_max3_f32, which LLVM's
    hazard recogniser does not cover. The fix is structural (mfma_settle in scan.hip: every accumulator tied through one asm
    statement of 20 wait states); this test holds it in the compiled code: in every filtering k_scan instantiation, each
    inline-asm v_max3_f32 has the settle statement between itself and the nearest MFMA above it."""
    import subprocess
    out = tmp_path / "scan.s"
    src = os.path.join(ROOT, "archi_amd", "csrc")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                    "-fno-fast-math", "-Wno-inline-asm", "-Wno-unused-command-line-argument", "--cuda-device-only", "-S",
                    "scan.hip", "-o", str(out)], cwd=src, check=True)
    fn, funcs = None, {}
    for ln in open(out):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            fn = m.group(1); funcs[fn] = []
        elif fn and ln.startswith("\t") and not ln.startswith("\t."):
            funcs[fn].append(ln.strip())
    checked = 0
    for name, ins in funcs.items():
        if "k_scan" not in name:
            continue
        in_asm, last_mfma, last_settle, n_asm_max3 = False, -1, -1, 0
        for i, x in enumerate(ins):
            if x.startswith(";;#ASMSTART"):
                in_asm = True
            elif x.startswith(";;#ASMEND"):
                in_asm = False
            elif x.startswith("v_mfma"):
                last_mfma = i
            elif in_asm and x.startswith("s_nop 15") and ins[i + 1].startswith("s_nop 3"):
                last_settle = i
            elif in_asm and x.startswith("v_max3_f32"):
                n_asm_max3 += 1
                assert last_settle > last_mfma >= 0, f"{name}: inline-asm v_max3 at {i} not behind mfma_settle"
        checked += n_asm_max3 > 0
    assert checked >= 12, checked          # 6 tiles x 2 dtypes of the main / seeding pass at least
