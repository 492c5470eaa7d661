"""GPU suite, early on purpose (fresh child processes; this process has not touched the GPU yet): every kernel-selection
switch of the encoder (A/B variants kept beside the launched kernels) must give the launched path's embeddings to bf16
noise -- the alternates are measurement tools and fallbacks, they may not rot.
  AK_ATTN_STREAM=0/1/2  k_attn / k_attn_s / k_attn_d at every head size      AK_QKV_GEMM=1  generic GEMM for the QKV projection
  AK_QKV_TG=1         16 tokens per wave in k_qkv384                           AK_FFN_ATT=0   out-projection in its own launch
  AK_FFN_W8=0         4-wave feed-forward kernel                               AK_ENC_NOFUSE=1 / AK_ENC_NOFFN=1  unfused hidden-384 path
  AK_FFN_NWV=4 / 8    64- / 128-token tiles of the fused layer kernel at every token count. 8 = the role-split kernel k_ffn384r with
                      its GELU read from the LDS table; AK_FFN_GELU=poly keeps the polynomial GELU in it, AK_FFN_ROLE=0 selects the
                      wave-pair kernel k_ffn384p and AK_FFN_PAIR=0 on top its predecessor k_ffn384w8 -- those three add every product
                      in the same order and must agree BIT FOR BIT
  AK_QK_TOKEN_MAJOR=1 q / k of the hidden-384 path as [T][384] rows instead of head-major (must equal the default BIT FOR BIT)
  AK_GEMM_BN=256 / 128, AK_GEMM_PHASED=0  the wide GEMM tile with the phased K-loop / the narrow tile / the wide tile's in-step loop
  AK_ENC_SKINNY_MAX=0 / 100000  128-token-tile kernels / small-batch kernels at every token count (the launched path switches
                      between them at 4096 tokens for hidden 384, 640 otherwise)
  AK_ENC_LAZYLN=2 / 0 lazy LayerNorm of the hidden-768 path (raw rows + per-token sums between the sub-layers, the LayerNorm folded
                      into the neighbouring GEMMs' epilogues; launched from ~11k tokens on) at every token count / off
"""
import concurrent.futures
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
VARIANTS = [{"AK_ATTN_STREAM": "2"}, {"AK_ATTN_STREAM": "1"}, {"AK_ATTN_STREAM": "0"}, {"AK_QKV_GEMM": "1"}, {"AK_QKV_TG": "1"}, {"AK_FFN_ATT": "0"},
            {"AK_FFN_W8": "0"}, {"AK_ENC_NOFFN": "1"}, {"AK_ENC_NOFUSE": "1"}, {"AK_ENC_SKINNY_MAX": "0"}, {"AK_ENC_SKINNY_MAX": "100000"},
            {"AK_FFN_NWV": "4", "AK_ENC_SKINNY_MAX": "0"}, {"AK_FFN_NWV": "8", "AK_ENC_SKINNY_MAX": "0"},
            {"AK_FFN_NWV": "8", "AK_ENC_SKINNY_MAX": "0", "AK_FFN_GELU": "poly"},
            {"AK_FFN_NWV": "8", "AK_ENC_SKINNY_MAX": "0", "AK_FFN_ROLE": "0"},
            {"AK_FFN_NWV": "8", "AK_ENC_SKINNY_MAX": "0", "AK_FFN_ROLE": "0", "AK_FFN_PAIR": "0"},
            {"AK_GEMM_BN": "256", "AK_ENC_SKINNY_MAX": "0"}, {"AK_GEMM_BN": "256", "AK_GEMM_PHASED": "0", "AK_ENC_SKINNY_MAX": "0"},
            {"AK_GEMM_BN": "128"}, {"AK_ENC_LAZYLN": "2", "AK_ENC_SKINNY_MAX": "0"}, {"AK_ENC_LAZYLN": "0"}]


# The child processes are independent (own HIP context, own output file): up to PAR of them share the GPU at a time, each with
# its share of the host cores for the torch-fp32 oracle. Every comparison is still made, only the waiting overlaps.
PAR = 3
_POOL = concurrent.futures.ThreadPoolExecutor(max_workers=PAR)


def _child_env(extra):
    env = {k: v for k, v in os.environ.items() if not k.startswith("AK_")}
    env.setdefault("OMP_NUM_THREADS", str(max(2, (os.cpu_count() or 8) // PAR)))
    env.update(extra)
    return env


def _run(tmp_path, name, extra):
    out = str(tmp_path / f"{name}.npz")
    env = _child_env(extra)
    p = subprocess.run([sys.executable, os.path.join(HERE, "encoder_variants_worker.py"), out], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode("utf-8", "replace")[-3000:]
    return np.load(out)


def test_kernel_selection_variants_agree(tmp_path):
    first = [_POOL.submit(_run, tmp_path, name, {}) for name in ("base", "again")]
    base, again = first[0].result(), first[1].result()
    for k in base.files:                              # the launched path is deterministic from process to process
        assert np.array_equal(base[k], again[k]), k
    runs = [_POOL.submit(_run, tmp_path, f"v{i}", extra) for i, extra in enumerate(VARIANTS)]
    for extra, fut in zip(VARIANTS, runs):
        got = fut.result()
        for k in base.files:
            cos = (got[k] * base[k]).sum(1)
            assert cos.min() >= 1 - 1e-4, (extra, k, float(cos.min()))
            assert np.abs(got[k] - base[k]).max() <= 2e-3, (extra, k)


def test_layer_kernels_with_the_polynomial_gelu_are_bit_identical(tmp_path):
    """k_ffn384r (producer / consumer waves) and k_ffn384p (wave pairs) split the feed-forward chunks differently but add every
    product in the order k_ffn384w8 does: with the same GELU (AK_FFN_GELU=poly) all three give the same bits. The launched
    kernel reads its GELU from the LDS table instead: held to the others at bf16 noise by the variant test above, and to the
    torch-fp32 oracle by test_oracle_comparisons_on_both_gemm_paths."""
    common = {"AK_FFN_NWV": "8", "AK_ENC_SKINNY_MAX": "0"}
    r = _run(tmp_path, "role", dict(common, AK_FFN_GELU="poly"))
    a = _run(tmp_path, "pair", dict(common, AK_FFN_ROLE="0"))
    b = _run(tmp_path, "w8", dict(common, AK_FFN_ROLE="0", AK_FFN_PAIR="0"))
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(r[k], b[k]), k


def test_head_major_q_k_layout_is_bit_identical_to_token_major(tmp_path):
    """k_qkv384 writes q / k as [B][heads][S][32] (whole cache lines for the attention kernel's staging); AK_QK_TOKEN_MAJOR=1
    keeps the [T][384] rows of the GEMM epilogue. A layout, not arithmetic: same bits."""
    common = {"AK_ENC_SKINNY_MAX": "0"}
    a = _run(tmp_path, "hm", common)
    b = _run(tmp_path, "tm", dict(common, AK_QK_TOKEN_MAJOR="1"))
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


ORACLE_VARIANTS = [{"AK_ENC_SKINNY_MAX": "0"}, {"AK_ENC_SKINNY_MAX": "100000"},
                   {"AK_ENC_SKINNY_MAX": "0", "AK_FFN_NWV": "8"},
                   {"AK_ENC_SKINNY_MAX": "0", "AK_GEMM_BN": "256"},
                   {"AK_ENC_SKINNY_MAX": "0", "AK_GEMM_BN": "256", "AK_GEMM_PHASED": "0"},
                   {"AK_ENC_SKINNY_MAX": "0", "AK_ENC_LAZYLN": "2"}]
_ORACLE_RUNS = {}


def _oracle_suite(extra):
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_encoder_gpu.py"), "-x", "-q", "-m", "gpu", "-k",
                        "hf_fixture or oracle or bge_base"], env=_child_env(extra), cwd=os.path.dirname(HERE),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
    return p.returncode, p.stdout.decode("utf-8", "replace"), p.stderr.decode("utf-8", "replace")


@pytest.mark.parametrize("extra", ORACLE_VARIANTS, ids=lambda e: ",".join(f"{k[3:]}={v}" for k, v in e.items()))
def test_oracle_comparisons_on_both_gemm_paths(extra):
    """The oracle / fixture comparisons of tests/test_encoder_gpu.py with the 128-token-tile kernels forced for every batch
    (AK_ENC_SKINNY_MAX=0) and with the small-batch kernels forced (100000): the suite's own batches are small, so the
    launched path alone would leave the tile kernels to a handful of cases. AK_GEMM_BN=256 on top forces the WIDE GEMM tile
    (256 features x 256 tokens; launched only from ~22k tokens on: the bench's 65 536-token batches) with its phased K-loop,
    and with the in-step loop it replaced (AK_GEMM_PHASED=0), for every hidden-768 GEMM of the suite. AK_FFN_NWV=8 puts every
    hidden-384 batch through the launched 128-token layer kernel (k_ffn384r, GELU by table), which small batches never reach.
    AK_ENC_LAZYLN=2 runs every hidden-768 batch through the lazy-LayerNorm GEMMs (launched from ~11k tokens on)."""
    if not _ORACLE_RUNS:                                # the first of these tests starts all of them, PAR at a time
        for e in ORACLE_VARIANTS:
            _ORACLE_RUNS[tuple(sorted(e.items()))] = _POOL.submit(_oracle_suite, e)
    rc, out, err = _ORACLE_RUNS[tuple(sorted(extra.items()))].result()
    assert rc == 0, out[-3000:] + err[-2000:]


@pytest.mark.parametrize("extra", [{"AK_X3_TILES": "2", "AK_X3_PADN": "0"}, {"AK_X3_TILES": "2", "AK_GEMM_BN": "256", "AK_X3_PADN": "15", "AK_X3_GEMMLN": "0"}, {"AK_X3_TILES": "0"}],
                         ids=lambda e: ",".join(f"{k[3:]}={v}" for k, v in e.items()))
def test_split_bf16_mode_on_both_gemm_families(extra):
    """precision="bf16x3" runs batches of >= 16 384 / 20 480 tokens (hidden 768 / 384) on gemm.hip's LDS-DMA tiles (operands as bf16 [hi | lo] rows, the K-loop
    walking 3 K: MODE 5 / 6) and smaller ones on encoder_f32.hip's k3_gemm. The suite's batches are small: AK_X3_TILES=2 puts every
    one of them on the tiles (narrow 128-feature tile; with AK_GEMM_BN=256 the wide phased tile, and with AK_X3_PADN=15 the output
    widths that are not multiples of 256 -- hidden 384 / 128 -- padded to one as large batches have them; AK_X3_GEMMLN=0: hidden 384 without the fused LayerNorm launches, i.e. MODE 5 + k3_add_ln
    as the other widths run), 0 keeps k3_gemm for all. Same
    bar either way: 1e-5 against transformers.BertModel / the float32 oracle."""
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_encoder_gpu.py"), "-x", "-q", "-m", "gpu", "-k",
                        "bf16x3 or split_bf16"], env=_child_env(extra), cwd=os.path.dirname(HERE),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert p.returncode == 0, p.stdout.decode("utf-8", "replace")[-3000:] + p.stderr.decode("utf-8", "replace")[-2000:]
