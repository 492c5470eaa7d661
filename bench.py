#!/usr/bin/env python3
"""bench.py -- kNN queries/sec @ top-10 on a 10M x 768 bf16 synthetic corpus (BASELINE.json
configs[2]), row-sharded over N MI355X of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one batch of Q queries (default 1024) searched against the whole corpus: MFMA
candidate scan with fused top-k' filter -> exact re-rank in the reference (pgvector) arithmetic ->
certificate -> (queries the scan could not certify are re-run exactly; none on this corpus); with
N>1 ranks: one RCCL all-gather of the per-shard partial top-k + certificate flags, merge kernel.
The corpus and the query batch are resident in HBM before the timed region starts; every step ends with its
ids + float8 distances copied to (pinned) host memory inside the timed region (SURVEY 8d's protocol).

The JSON line carries
  roofline     : the scan kernel (k_scan) against the bf16 MFMA peak (Q >= 320) or HBM peak,
                 achieved = algorithmic flops|bytes per launch / mean launch duration measured
                 live with HIP events recorded around that kernel on the launch stream.
  cpu_baseline : the CPU port of the reference path timed on this box's host cores on a bounded
                 sample of the same workload (rank 0, N=1 only).
  verified     : what was CHECKED in this run, outside the timed region (the run exits non-zero on
                 a mismatch): returned rows re-scored by the oracle, the exact HIP path, the
                 oracle's own scan of a row slice, and at N>1 the single-index result.
  pcie_inclusive / step_ms_hip_events : SURVEY 8d's protocol fields (never `value`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFS = 2500.0  # same guide: ~2.5 PF dense bf16/f16 MFMA
MFMA_F32_PEAK_TFS = 157.3    # same guide: v_mfma_f32_32x32x2_f32, 64 FLOP / clk / SIMD (= the fp32 vector peak)
PEAK_SCLK_MHZ = 2400.0       # the engine clock the guide's matrix peaks are quoted at


def add_clock_adjusted(roof):
    """Beside `frac` (against the NOMINAL peak, the contract's number): what the matrix roof is at the clock the part sustained
    under THIS load -- every leg of this bench runs into the 1.4 kW board power cap and the shader clock drops to 1.7-2.15 GHz
    (round 5 telemetry) -- and the fraction of that. MFMA-bound rooflines only; None without telemetry."""
    sclk = roof.get("sclk_mhz_under_load")
    if roof.get("bound") == "mfma" and sclk:
        roof["peak_at_sustained_clock"] = roof["peak"] * min(1.0, sclk / PEAK_SCLK_MHZ)
        roof["frac_of_peak_at_sustained_clock"] = roof["achieved"] / roof["peak_at_sustained_clock"]
    else:
        roof["peak_at_sustained_clock"] = roof["frac_of_peak_at_sustained_clock"] = None
    return roof


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-embed", action="store_true", help="skip the chunk-embeds/sec leg")
    ap.add_argument("--comm", default="torch", choices=["torch", "abi"],
                    help="exchange step of the row-sharded search: torch.distributed all_gather_into_tensor (default) or the "
                         "C-ABI path (ak_index_search_sharded_dev: ncclAllGather issued by libarchi_hip.so itself)")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="skip hbm_bound_configs (profiling runs: keeps the kernel trace to the main workload's launches)")
    ap.add_argument("--embed-batch", type=int, default=256)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU baseline budget")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the in-run result verification (outside the timed region; on by default): sampled queries "
                         "re-scored by the oracle, compared with the exact HIP path and with the oracle's scan of a row "
                         "slice; with N > 1 ranks, rank 0 also builds the WHOLE corpus as one index and checks that the "
                         "sharded result (all-gather + merge) is identical")
    return ap.parse_args()


def vendor_gemm_tflops(n=8192, reps=10):
    """What the vendor's tuned GEMM (torch.matmul -> hipBLASLt, bf16, n^3) sustains on THIS GPU, measured in the same
    run: context for roofline.frac, which is quoted against the nominal 2.5 PFLOP/s. Not part of the timed region."""
    a = torch.randn(n, n, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(n, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        a @ b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        a @ b
    torch.cuda.synchronize()
    return 2.0 * n ** 3 * reps / (time.perf_counter() - t0) / 1e12


class GpuTelemetry:
    """Shader clock and board power WHILE a timed loop runs, so that a reader of the JSON line can tell a slow box from slow code
    (round-4 review: the same kernels ran 9 % apart on two boxes). A background thread samples the amdgpu sysfs files of the
    bound device (hwmon freq1_input = sclk in Hz, power1_average / power1_cap in microwatts; fallback: one `rocm-smi` call per
    sample); it touches neither the GPU runtime nor the timed stream. Fields stay None where the box exposes nothing."""

    def __init__(self, device_index=0, period_s=0.05, sysfs_root="/sys/class/drm", pci=None):
        import glob
        import threading
        self.period, self.samples, self.stop_flag = period_s, [], threading.Event()
        self.hw = None
        cards = sorted(glob.glob(os.path.join(sysfs_root, "card*/device/hwmon/hwmon*")))
        cards = [c for c in cards if os.path.exists(os.path.join(c, "freq1_input"))]
        # The box's sysfs lists every GPU of the host, the process sees one: the card is the one whose PCI address is the bound
        # device's (card*/device -> ../../../dddd:bb:dd.f). Without a match there is no telemetry (round 5, first visits: card0 was
        # somebody else's GPU -- 158 MHz / 269 W through a whole scan loop).
        self.pci = pci
        if self.pci is None:
            try:
                pr = torch.cuda.get_device_properties(device_index)
                self.pci = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            except Exception:
                pass
        if self.pci:
            cards = [c for c in cards if os.path.basename(os.path.realpath(c.split("/hwmon/")[0])).lower().startswith(self.pci)]
            self.hw = cards[0] if cards else None
        elif len(cards) == 1:
            self.hw = cards[0]
        # rocm-smi numbers the host's GPUs too: only where sysfs shows no amdgpu hwmon at all
        self.smi = None if (self.hw or cards or self.pci) else next((p for p in ("/opt/rocm/bin/rocm-smi", "/usr/bin/rocm-smi") if os.path.exists(p)), None)
        self.dev = device_index
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _read(self, name, scale):
        try:
            return float(open(os.path.join(self.hw, name)).read().strip()) * scale
        except (OSError, ValueError):
            return None

    def _sample(self):
        if self.hw:
            return (self._read("freq1_input", 1e-6), self._read("power1_average", 1e-6) or self._read("power1_input", 1e-6),
                    self._read("power1_cap", 1e-6))
        if self.smi:
            import subprocess
            try:
                o = subprocess.run([self.smi, "-d", str(self.dev), "--showclocks", "--showpower", "--showmaxpower", "--json"],
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=5).stdout.decode()
                d = list(json.loads(o).values())[0]
                sclk = next((float(str(v).strip("()MmHhz ")) for k, v in d.items() if "sclk clock speed" in k.lower()), None)
                pw = next((float(v) for k, v in d.items() if "power (w)" in k.lower() and "max" not in k.lower()), None)
                cap = next((float(v) for k, v in d.items() if "max graphics package power" in k.lower()), None)
                return (sclk, pw, cap)
            except Exception:
                return None
        return None

    def _run(self):
        while not self.stop_flag.is_set():
            s = self._sample()
            if s:
                self.samples.append(s)
            self.stop_flag.wait(self.period if self.hw else max(self.period, 0.5))

    def __enter__(self):
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop_flag.set()
        self.thread.join(timeout=10)

    def summary(self):
        def med(i):
            v = [s[i] for s in self.samples if s[i] is not None]
            return float(np.median(v)) if v else None
        return {"sclk_mhz_under_load": med(0), "power_w_under_load": med(1), "power_cap_w": med(2),
                "telemetry_samples": len(self.samples),
                "telemetry_source": self.hw or (self.smi and "rocm-smi") or None, "telemetry_pci": self.pci}


def vendor_knn_qps(nq, dim, k, total_rows, slice_rows=1_000_000, reps=5):
    """Context: brute-force top-k through PyTorch-ROCm's own kernels on this GPU -- bf16 matmul (hipBLASLt) + torch.topk
    over a slice, scaled linearly to the corpus (the scan is O(rows); merging the per-slice top-k is not even charged).
    Approximate scores, no exact re-rank: an upper bound on what that stack does for this workload."""
    rows = torch.randn(slice_rows, dim, device="cuda", dtype=torch.bfloat16)
    q = torch.randn(nq, dim, device="cuda", dtype=torch.bfloat16)
    for _ in range(2):
        torch.topk(q @ rows.T, k, dim=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        torch.topk(q @ rows.T, k, dim=1)
    torch.cuda.synchronize()
    per_slice = (time.perf_counter() - t0) / reps
    return nq / (per_slice * total_rows / slice_rows)


def gen_queries(nq, dim, dtype, batch=0):
    """Synthetic query batch from the same counter-based generator (stream 1), via the product API."""
    from archi_amd.index import HipIndex
    tmp = HipIndex(dim, nq, dtype=dtype, metric="cosine")
    tmp.generate(seed=4321, n=nq, stream=1, normalise=True, row0=batch * nq)
    q = tmp.fetch(np.arange(nq))
    tmp.close()
    return q


def hbm_bound_configs(ix, args):
    """The HBM-bound members of BASELINE.json's configs, outside the timed region: small query batches on
    the resident corpus (cfg3 Q in {1, 64}) and cfg2 (1M x 384 fp32 rows, Q in {1, 16, 256}). Per point: whole-search
    time (device-resident, 20 back-to-back searches) and the scan launch alone (HIP events) against 8 TB/s."""
    from archi_amd.index import HipIndex
    from archi_amd.sharded import HipLocalSearch
    out = []

    def point(index, label, rows, dim, nq, stream_bytes_per_row, note):
        q = torch.from_numpy(gen_queries(nq, dim, args.dtype if index is ix else "f32")).cuda()
        loc = HipLocalSearch(index)
        for _ in range(3):
            loc(q, args.k)
        torch.cuda.synchronize()
        index.profile(True)
        t0 = time.perf_counter()
        for _ in range(20):
            loc(q, args.k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        scan_ms = index.profile_read()
        index.profile(False)
        plan = index.scan_plan(nq, args.k)
        main_rows = rows - plan["seed_rows"]
        gbs = main_rows * stream_bytes_per_row / (float(scan_ms.mean()) * 1e-3) / 1e9 if scan_ms.size else None
        tfs = 2.0 * nq * main_rows * dim / (float(scan_ms.mean()) * 1e-3) / 1e12 if scan_ms.size else None
        cert_n = int(loc.last_cert.sum().item())
        out.append({"config": label, "queries": nq, "search_ms": ms, "queries_per_s": nq / ms * 1e3,
                    "scan_launch_ms": float(scan_ms.mean()) if scan_ms.size else None, "scan_tile": plan["cfg_name"],
                    "scan_GB_per_s": gbs, "frac_of_8TBps": gbs / HBM_PEAK_GBS if gbs else None,
                    "scan_TFLOP_per_s": tfs, "frac_of_mfma_peak": tfs / MFMA_BF16_PEAK_TFS if tfs else None,
                    "certified": cert_n, "note": note})

    for nq in (1, 64, 256, 384):       # 256 / 384: the ridge regime (neither roof binds; tile chosen by (Q, N, D))
        point(ix, f"cfg3 {args.rows}x{args.dim} {args.dtype}", args.rows, args.dim, nq, args.dim * 2, "rows streamed once, 2 B per element")
    c2 = HipIndex(384, 1_000_000, dtype="f32", metric="cosine")
    c2.generate(seed=1234, n=1_000_000, stream=0, normalise=True)
    for nq in (1, 16, 256):
        point(c2, "cfg2 1000000x384 f32", 1_000_000, 384, nq, 384 * 2,
              "candidates from the bf16 shadow of the fp32 rows (2 B per element streamed), exact re-rank on the fp32 rows")
    c2.close()
    return out


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _torch_blas():
    """Which BLAS backs torch.mm in the CPU baseline (from torch's build configuration)."""
    try:
        cfg = torch.__config__.show()
        for key in ("BLAS_INFO=", "LAPACK_INFO="):
            if key in cfg:
                return cfg.split(key, 1)[1].split(",")[0].split()[0] + (" (+ oneDNN)" if "USE_MKLDNN=ON" in cfg or "USE_MKLDNN=1" in cfg else "")
    except Exception:
        pass
    return "unknown"


def _physical_cores():
    """Distinct (physical id, core id) pairs of /proc/cpuinfo (SMT siblings counted once); falls back to the CPU count."""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


def _cpu_worker(spec):
    """Hidden mode `--cpu-worker threads:first_cpu:nq:dim:rows:seconds:k` (a child process of cpu_baseline, never touches
    the GPU): the B2 loop -- fp32 sgemm of the query batch against its own row slice + top-k -- on `threads` threads pinned
    to CPUs [first_cpu, first_cpu + threads); prints query-rows per second."""
    parts = spec.split(":")
    th, first, nq, dim, rows, secs, k = parts[:7]
    blas = parts[7] if len(parts) > 7 else "torch"
    th, first, nq, dim, rows, k, secs = int(th), int(first), int(nq), int(dim), int(rows), int(k), float(secs)
    try:                                       # the first..first+th-th CPUs this process is ALLOWED on (a cpuset may not start at 0)
        allowed = sorted(os.sched_getaffinity(0))
        if len(allowed) >= first + th:
            os.sched_setaffinity(0, set(allowed[first:first + th]))
    except (OSError, ValueError, AttributeError):
        pass
    torch.set_num_threads(th)
    g = torch.Generator().manual_seed(first)
    q = torch.randn(nq, dim, generator=g)
    r = torch.randn(rows, dim, generator=g)
    if blas in ("numpy", "numpy-zen"):          # "numpy-zen": the same with OPENBLAS_CORETYPE=ZEN in the worker's environment (set by the parent)
        # numpy's bundled OpenBLAS (AVX-512 kernels for Zen; torch's MKL takes a slower path on AMD hosts -- round-4 review):
        # OpenBLAS sgemm, then torch.topk on the scores in place (np.argpartition is single-threaded: 5x the sgemm's time here)
        qn, rn = q.numpy(), np.ascontiguousarray(r.numpy())

        def one():
            torch.topk(torch.from_numpy(qn @ rn.T), k, dim=1)
    else:
        def one():
            torch.topk(q @ r.T, k, dim=1)
    one()
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < secs:
        one(); reps += 1
    print(json.dumps({"qrows_per_s": nq * rows * reps / (time.perf_counter() - t0), "reps": reps}), flush=True)


def run_cpu_workers(cmds, env, deadline_s):
    """Start one child per command, sum the `qrows_per_s` of their last JSON lines. A worker that does not come back within
    `deadline_s` (round 5, one box: a 4-thread worker pinned to CPUs 0-3 sat for 120 s and the exception took the whole bench
    down) ends the probe: every child is killed and reaped, the layout counts as NaN, the caller's sweep goes on."""
    import subprocess
    procs = [subprocess.Popen(c, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for c in cmds]
    tot, deadline = 0.0, time.perf_counter() + deadline_s
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=max(1.0, deadline - time.perf_counter()))
            tot += json.loads(o.decode().strip().splitlines()[-1])["qrows_per_s"]
        except subprocess.TimeoutExpired:
            for q_ in procs:
                if q_.poll() is None:
                    q_.kill()
            for q_ in procs:
                try:
                    q_.communicate(timeout=10)
                except Exception:
                    pass
            return float("nan")
        except Exception:
            tot = float("nan")
    return tot


def cpu_baseline(ix, queries, k, total_rows, budget_s):
    """CPU port of the reference read path on a bounded sample of the same workload.

    B2 'best-effort CPU' (the stronger baseline, reported as `value`): batched fp32 sgemm of the query batch against the
    up-cast rows + top-k (torch.mm + torch.topk), the corpus rows partitioned over P worker processes of T threads each,
    every worker pinned to its own block of CPUs. One process with all the threads in a single GEMM is one of the layouts
    tried (P = 1), but on a many-CCD host it is far from the best: the layout (P x T) is the winner of a sweep whose probes
    run >= 2.5 s each, all layouts reported.
    B1 'pgvector-faithful' (reported beside it): the oracle's sequential float32 scan + heap, one query at a time on one
    core -- what one Postgres backend does on the exact-scan branch (/root/reference/src/cli/templates/init.sql:290-292).
    Both are timed on row slices and scaled linearly to the full corpus (the scan is O(rows))."""
    import subprocess
    from oracle import knn_oracle as ko
    try:
        cpus = len(os.sched_getaffinity(0)) or 1
    except (OSError, AttributeError):
        cpus = os.cpu_count() or 1
    phys = min(_physical_cores(), cpus)
    nq, dim = queries.shape
    layouts = []
    for p_, t_ in ((1, 32), (1, phys), (4, 32), (8, 16), (16, 8), (32, 4), (64, 2), (32, 8), (64, 4), (cpus // 16 or 1, 16)):
        if p_ >= 1 and t_ >= 1 and p_ * t_ <= cpus and (p_, t_) not in layouts:
            layouts.append((p_, t_))
    probe_s = max(2.5, min(4.0, budget_s / max(len(layouts), 1) - 2.0))
    rows_w = 65536                                          # rows per worker slice: 1024 x 65536 x 768 = 0.1 TFLOP per repetition

    def probe(p_, t_, blas):
        env = dict(os.environ, OMP_NUM_THREADS=str(t_), MKL_NUM_THREADS=str(t_), OPENBLAS_NUM_THREADS=str(t_))
        if blas == "numpy-zen":                   # round-5 review: OpenBLAS's runtime dispatch may not pick its Zen kernels on a Zen 5 host
            env["OPENBLAS_CORETYPE"] = "ZEN"
        cmds = [[sys.executable, os.path.abspath(__file__), "--cpu-worker", f"{t_}:{i * t_}:{nq}:{dim}:{rows_w}:{probe_s}:{k}:{blas}"]
                for i in range(p_)]
        return run_cpu_workers(cmds, env, 8.0 * probe_s + 20.0)     # a probe is one warm-up repetition + probe_s seconds
    sweep_t0 = time.perf_counter()

    def probe_bounded(p_, t_, blas):              # the whole sweep stays inside ~3x its budget whatever the host does
        return probe(p_, t_, blas) if time.perf_counter() - sweep_t0 < 3.0 * budget_s + 60.0 else float("nan")
    sweep = {f"{p_}x{t_}": probe_bounded(p_, t_, "torch") for p_, t_ in layouts}
    good = {kk: v for kk, v in sweep.items() if v == v}
    if not good:
        raise RuntimeError("cpu_baseline: no layout of the sweep finished")
    best = max(good, key=good.get)
    # the other BLAS on the layouts that matter: torch's winner, one process with every core, and two many-process layouts
    np_layouts = []
    for lay in (best, f"1x{phys}", "8x16"):
        p_, t_ = (int(x) for x in lay.split("x"))
        if p_ * t_ <= cpus and lay not in np_layouts:
            np_layouts.append(lay)
    sweep_np = {lay: probe_bounded(*(int(x) for x in lay.split("x")), "numpy") for lay in np_layouts}
    # ... and OpenBLAS told to use its Zen kernels (OPENBLAS_CORETYPE=ZEN), with one worker per CCD-sized block of cores (8 / 16
    # threads, pinned): the layouts a Zen host's cache topology suggests
    for lay in ("16x8", "8x16"):
        p_, t_ = (int(x) for x in lay.split("x"))
        if p_ * t_ <= cpus:
            sweep_np[lay + " OPENBLAS_CORETYPE=ZEN"] = probe_bounded(p_, t_, "numpy-zen")
    good_np = {kk: v for kk, v in sweep_np.items() if v == v}
    best_np = max(good_np, key=good_np.get) if good_np else None
    winner_blas = "numpy (bundled OpenBLAS) sgemm + torch.topk" if best_np and good_np[best_np] > good[best] else _torch_blas() + " (torch.mm)"
    if best_np and good_np[best_np] > good[best]:
        if "CORETYPE" in best_np:
            winner_blas = "numpy (bundled OpenBLAS, OPENBLAS_CORETYPE=ZEN) sgemm + torch.topk"
        best, top = best_np, good_np[best_np]
    else:
        top = good[best]
    b2_qps = top / total_rows                              # query-rows per second / rows per query
    bp, bt = (int(x) for x in best.split(" ")[0].split("x"))
    try:
        ghz = float(open("/sys/devices/system/cpu/cpu0/cpufreq/cpuinfo_max_freq").read()) / 1e6
    except (OSError, ValueError):
        ghz = None
    # B1: oracle C, single thread, a few queries on a slice of the stored rows
    s1 = int(min(100_000, ix.slots))
    nq1 = min(4, nq)
    rows = ix.fetch(np.arange(s1))                          # stored values up-cast to fp32
    t0 = time.perf_counter()
    oi, od, _ = ko.search(rows, queries[:nq1], k, "cosine")
    b1_s = time.perf_counter() - t0
    b1_qps = nq1 / (b1_s * total_rows / s1)
    # sanity: the B2 arithmetic's top-1 agrees with the oracle on the shared slice
    chk = torch.topk(torch.from_numpy(queries[:nq1]) @ torch.from_numpy(rows).T, 1, dim=1).indices[:, 0].numpy()
    agree = bool((chk == oi[:, 0]).all())
    return {
        "value": b2_qps, "unit": "queries/s", "cores": bp * bt, "kind": "port",
        "cores_note": f"{bp} worker processes x {bt} threads = {bp * bt} threads (the winner of the layout sweep) on "
                      f"{phys} physical cores / {cpus} logical CPUs"
                      + (": one thread per physical core" if bp * bt == phys else
                         ": one thread per logical CPU (SMT)" if bp * bt == cpus else ""),
        "blas": winner_blas,
        "achieved_sgemm_gflops": round(2.0 * dim * top / 1e9, 1),
        "host_fp32_peak_gflops_nominal": (round(phys * 64 * ghz, 0) if ghz else None),
        "host_peak_note": "physical cores x 64 FLOP / cycle (two 512-bit FMA pipes) x cpuinfo_max_freq: a nominal figure, AVX-512 "
                          "clocks run lower; the baseline is a reported context number, the kernel's roofline fraction is the grade",
        "layout_sweep_numpy_openblas_sgemm_gflops": {kk: (round(2.0 * dim * v / 1e9, 1) if v == v else None) for kk, v in sweep_np.items()},
        "host_cpus": cpus, "physical_cores": phys, "cpu_model": _cpu_model(),
        "sample": f"B2 fp32 sgemm+topk, {bp} worker processes x {bt} threads (winner of a layout sweep, {probe_s:.1f} s per probe): "
                  f"{nq} queries x {rows_w} rows per worker per repetition, scaled linearly to {total_rows} rows; "
                  f"B1 oracle C scan: {nq1} queries x {s1} rows on 1 core",
        "layout_sweep_qps": {kk: (v / total_rows if v == v else None) for kk, v in sweep.items()},
        "layout_sweep_sgemm_gflops": {kk: (round(2.0 * dim * v / 1e9, 1) if v == v else None) for kk, v in sweep.items()},
        "b1_pgvector_faithful_qps_1core": b1_qps, "b2_top1_agrees_with_oracle": agree,
    }


def verify_results(args, full_ix, q_host, ids, dd, k, slice_rows=200_000, n_sample=8):
    """Outside the timed region: check what the timed loop returned (ids, dd: numpy [Q,k] of the LAST timed step).
    (1) the returned rows of `n_sample` sampled queries, regenerated by the oracle's generator, re-score to the same
        float8 bits in the oracle's arithmetic;
    (2) the same queries through the exact HIP path (reference arithmetic for every row) give the same rows;
    (3) the same queries restricted to rows [0, slice_rows) (WHERE mask) equal the oracle's own scan (B1) of that slice.
    `full_ix` holds the whole corpus. Returns the `verified` object; raises SystemExit on any mismatch."""
    from oracle import knn_oracle as ko
    nq = q_host.shape[0]
    pick = sorted(set(np.linspace(0, nq - 1, min(n_sample, nq)).astype(int).tolist()))
    rescored = 0
    for qi in pick:
        for j in range(k):
            row = ko.gen_rows(1234, 0, int(ids[qi, j]), 1, args.dim, True, args.dtype)[0]
            if ko.distance("cosine", row, q_host[qi]) != dd[qi, j]:
                raise SystemExit(f"bench.py verify: query {qi} rank {j}: oracle re-score differs from the returned distance")
            rescored += 1
    qs = np.ascontiguousarray(q_host[pick])
    ei, ed, _ = full_ix.search(qs, k, mode="exact")
    if not (np.array_equal(ei, ids[pick]) and np.array_equal(ed, dd[pick])):
        raise SystemExit("bench.py verify: the timed path's rows differ from the exact HIP path")
    s_rows = int(min(slice_rows, args.rows))
    flt = np.zeros(full_ix.slots, np.uint8); flt[:s_rows] = 1
    gi, gd, _ = full_ix.search(qs, k, mode="auto", row_filter=flt)
    corpus = ko.gen_rows(1234, 0, 0, s_rows, args.dim, True, args.dtype)
    oi, od, _ = ko.search(corpus, qs, k, "cosine")
    if not (np.array_equal(gi, oi) and np.array_equal(gd, od)):
        raise SystemExit("bench.py verify: HIP search of the row slice differs from the oracle's scan of it")
    return {"sampled_queries": pick, "rows_rescored_by_oracle": rescored, "equals_exact_hip_path": True,
            "equals_oracle_scan_of_slice": True, "slice_rows": s_rows,
            "what": "ids and float8 distance bits of the last timed step, compared with array_equal"}


def _ev_summary(step_ms):
    if not len(step_ms):
        return None
    a = np.asarray(step_ms)
    return {"median": float(np.median(a)), "min": float(a.min()), "max": float(a.max()), "n": int(a.size),
            "note": "per-step HIP events on the launch stream (this rank); `value` is wall clock over all steps, max over ranks"}


def timed_loop(step, world, min_steps, min_seconds=0.5, warm_seconds=0.5, warm_cap_seconds=4.0):
    """Secondary legs (the encoder shapes): warm up BY TIME, then time enough steps that the window is not a blip.

    Warm-up: at least `warm_seconds` of back-to-back steps, continued (up to `warm_cap_seconds`) until three consecutive
    HIP-event step times agree within 3 % -- a leg that starts after tens of seconds of GPU idle (the CPU baseline runs
    before it) meets a down-clocked part and first-use allocations, and a five-step warm-up measured that, not the kernels.
    Timed region: max(min_steps, enough for `min_seconds`) steps between synchronise + barrier on both sides; every step
    also carries a HIP event pair on the launch stream. Returns (wall seconds (max over ranks), steps, per-step ms, warm-up
    record)."""
    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    t_w = time.perf_counter()
    last, n_warm, stable = [], 0, False
    while True:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); step(); b.record(); b.synchronize()
        n_warm += 1
        last = (last + [a.elapsed_time(b)])[-3:]
        spent = time.perf_counter() - t_w
        stable = len(last) == 3 and max(last) <= 1.03 * min(last)
        if (spent >= warm_seconds and stable) or spent >= warm_cap_seconds:
            break
    warm_s = time.perf_counter() - t_w
    est = float(np.median(last)) * 1e-3
    n = int(max(min_steps, np.ceil(min_seconds / max(est, 1e-6))))
    if world > 1:                                   # every rank must run the same number of steps
        nt = torch.tensor([n], dtype=torch.int64, device="cuda")
        dist.all_reduce(nt, op=dist.ReduceOp.MAX)
        n = int(nt.item())
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    sync_all()
    t0 = time.perf_counter()
    for i in range(n):
        ev[i][0].record()
        step()
        ev[i][1].record()
    sync_all()
    el = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
    step_ms = [x.elapsed_time(y) for x, y in ev]
    return el, n, step_ms, {"steps": n_warm, "seconds": round(warm_s, 3), "stable_within_3pct": bool(stable)}


def embed_bench(args, world, rank, local_rank, with_cpu):
    """chunk-embeds/sec: the other half of BASELINE.json's metric. all-MiniLM-L6-v2 architecture
    (the reference default, src/cli/templates/base-config.yaml:145), seeded random-init weights,
    [B,256] synthetic token ids (uniform in [1000,30000), full mask). Pure data parallel over ranks
    (weights replicated, no collective on the data path)."""
    from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights
    name = "sentence-transformers/all-MiniLM-L6-v2"
    vocab, H, L, heads, I, max_pos, pooling, S = MODEL_SHAPES[name]
    B = args.embed_batch
    weights = random_init_weights(vocab, H, L, I, max_pos, seed=0)
    enc = HipEncoder(vocab, H, L, heads, I, max_pos, weights, device=local_rank)
    rng = np.random.default_rng(1000 + rank)
    ids_h = rng.integers(1000, 30000, size=(B, S)).astype(np.int32)
    ids = torch.from_numpy(ids_h).cuda()
    mask = torch.ones((B, S), dtype=torch.int32, device="cuda")
    box = {}

    def step():
        box["emb"] = enc.forward(ids, mask, pooling=pooling)

    with GpuTelemetry(local_rank) as tele:
        el, n_steps, step_ms, warm = timed_loop(step, world, min_steps=max(100, args.steps))
    emb = box["emb"]
    chunks_s = world * B * n_steps / el
    flops_chunk = S * L * (2 * (4 * H * H + 2 * H * I) + 4 * S * H)
    tfs = chunks_s / world * flops_chunk / 1e12
    res = {"metric": "chunk-embeds/sec (256-token chunks)", "value": chunks_s, "unit": "chunks/s",
           "ms_per_step": el * 1e3 / n_steps, "steps": n_steps, "warmup": warm, "dtype": "bf16",
           "step_ms_hip_events": _ev_summary(step_ms),
           "config": {"workload": f"all-MiniLM-L6-v2 architecture (6 layers, H=384, 12 heads, FFN 1536), random-init "
                                  f"seed 0, {B} x {S} synthetic token ids per rank, mean-pool + L2 normalise; bf16 MFMA inputs, "
                                  f"fp32 accumulate / LayerNorm / softmax, residual stream in {enc.residual}",
                      "parallelism": f"dp{world} (replicated weights, no collective)"},
           "roofline": {"bound": "mfma", "achieved": tfs, "peak": MFMA_BF16_PEAK_TFS, "unit": "TFLOP/s",
                        "frac": tfs / MFMA_BF16_PEAK_TFS, "algorithmic_flops_per_chunk": flops_chunk,
                        "note": "whole forward pass (all kernels), per GPU"}}
    res["roofline"].update(tele.summary())
    add_clock_adjusted(res["roofline"])
    vg = getattr(args, "vendor_gemm_tflops", None)
    res["roofline"]["vendor_gemm_tflops"] = vg
    res["roofline"]["achieved_over_vendor_gemm"] = (tfs / vg) if vg else None
    if rank == 0:
        # context, outside the timed region: the same architecture through PyTorch-ROCm's own stack
        # (transformers.BertModel, bf16, SDPA attention -> hipBLASLt / vendor kernels) on this GPU, same batch shape
        try:
            from transformers import BertConfig, BertModel
            cfg = BertConfig(vocab_size=vocab, hidden_size=H, num_hidden_layers=L, num_attention_heads=heads,
                             intermediate_size=I, max_position_embeddings=max_pos, hidden_act="gelu", layer_norm_eps=1e-12)
            hf = BertModel(cfg, add_pooling_layer=False).cuda().to(torch.bfloat16).eval()
            ids64 = ids.long()
            with torch.no_grad():
                for _ in range(3):
                    hf(input_ids=ids64, attention_mask=mask).last_hidden_state
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    hf(input_ids=ids64, attention_mask=mask).last_hidden_state
                torch.cuda.synchronize()
            hf_s = (time.perf_counter() - t0) / 10
            res["vendor_stack"] = {"what": "transformers.BertModel bf16 + SDPA on PyTorch-ROCm, same GPU, same [B,S] (no pooling)",
                                   "chunks_per_s": B / hf_s, "this_build_over_vendor_stack": (chunks_s / world) / (B / hf_s)}
            del hf
        except Exception as e:                      # context only
            res["vendor_stack"] = {"error": str(e)[:200]}
    if with_cpu and rank == 0:
        from oracle import encoder_oracle as eo
        torch.set_num_threads(min(os.cpu_count() or 1, 64))
        w = {k: (v if isinstance(v, np.ndarray) else np.asarray(v)) for k, v in weights.items()}
        nb, nbatches = 32, 3                      # torch-CPU fp32, batch 32, all cores
        nb = min(nb, B)
        eo.forward("minilm-l6", w, ids_h[:2], np.ones((2, S), np.int32))
        t0 = time.perf_counter()
        for _ in range(nbatches):
            ref = eo.forward("minilm-l6", w, ids_h[:nb], np.ones((nb, S), np.int32))
        dt = (time.perf_counter() - t0) / nbatches
        got = emb[:nb].cpu().numpy()
        cos = float(((got * ref).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(ref, axis=1))).min())
        res["cpu_baseline"] = {"value": nb / dt, "unit": "chunks/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"torch fp32 CPU restatement of the same forward pass, {nbatches} batches of {nb} of the {B} chunks",
                               "min_cosine_gpu_vs_cpu": cos}
    if with_cpu and rank == 0:
        # END TO END against the CPU path on the same inputs (north star: "same top-k as the reference CPU path"), outside the
        # timed region: 384 ragged synthetic chunks + 16 queries through (a) the torch-fp32 encoder oracle -> oracle cosine top-10,
        # (b) the HIP encoder in float32 parity mode, (b') in split-bf16 parity mode and (c) in the default bf16 mode -> float32 HipIndex -> top-10
        try:
            from archi_amd.index import HipIndex
            from oracle import encoder_oracle as eo
            from oracle import knn_oracle as ko
            e_rng = np.random.default_rng(77)
            n_c, n_q = 384, 16
            lens = np.concatenate([e_rng.integers(32, S + 1, size=n_c), e_rng.integers(8, 49, size=n_q)])
            toks = e_rng.integers(1000, 30000, size=(n_c + n_q, S)).astype(np.int32)
            msk = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
            toks *= msk
            wnp = {k: (v if isinstance(v, np.ndarray) else np.asarray(v)) for k, v in weights.items()}
            ref = np.concatenate([eo.forward("minilm-l6", wnp, toks[o:o + 32], msk[o:o + 32]) for o in range(0, n_c + n_q, 32)])
            ri, rd, _ = ko.search(ref[:n_c], ref[n_c:], 10, "cosine")
            e2e = {"what": f"{n_c} ragged synthetic chunks + {n_q} queries: torch-fp32 encoder oracle + oracle cosine top-10 (CPU path) "
                           "against the HIP encoder + float32 HipIndex"}
            for mode in ("f32", "bf16x3", "bf16"):
                e = enc if mode == "bf16" else HipEncoder(vocab, H, L, heads, I, max_pos, weights, device=local_rank, precision=mode)
                got = np.concatenate([e.forward(torch.from_numpy(toks[o:o + 128]).cuda(), torch.from_numpy(msk[o:o + 128]).cuda(),
                                                pooling=pooling).cpu().numpy() for o in range(0, n_c + n_q, 128)])
                ix = HipIndex(H, n_c, dtype="f32", metric="cosine", device=local_rank)
                ix.add(got[:n_c])
                gi, gd, _ = ix.search(got[n_c:], 10)
                ix.close()
                if mode != "bf16":
                    e.close()
                ov = [len(set(gi[j].tolist()) & set(ri[j].tolist())) / 10.0 for j in range(n_q)]
                dsc = max(abs((1.0 - gd[j, p]) - (1.0 - ko.distance("cosine", ref[int(gi[j, p])], ref[n_c + j])))
                          for j in range(n_q) for p in range(10))
                e2e[mode] = {"overlap_at_10_mean": float(np.mean(ov)), "overlap_at_10_min": float(min(ov)),
                             "ids_identical": bool(np.array_equal(gi, ri)), "max_abs_score_diff": float(dsc)}
            res["end_to_end"] = e2e
        except Exception as exc:                    # context only, never the metric
            res["end_to_end"] = {"error": str(exc)[:200]}
    if rank == 0:
        # SURVEY 8d's ragged-mask variant, outside the timed region: chunk lengths uniform in [32,S]. (a) the same
        # [B,S] tile with a ragged attention mask; (b) 4096 ragged chunks (token-id rows + lengths, what the host tokenizer emits) through the provider's length-sorted
        # [B',S'] tiling (host harness included: token lists in, float32 rows out on the host)
        try:
            lens = rng.integers(32, S + 1, size=B)
            rmask = torch.from_numpy((np.arange(S)[None, :] < lens[:, None]).astype(np.int32)).cuda()
            for _ in range(2):
                enc.forward(ids, rmask, pooling=pooling)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                enc.forward(ids, rmask, pooling=pooling)
            torch.cuda.synchronize()
            padded_s = (time.perf_counter() - t0) / 10
            from archi_amd.embeddings import ArchiHipEmbeddings
            prov = ArchiHipEmbeddings(name, model_kwargs={"synthetic_seed": 0, "device": f"cuda:{local_rank}"},
                                      encode_kwargs={"normalize_embeddings": True})
            n_r = 4096
            rl = rng.integers(32, S + 1, size=n_r).astype(np.int32)
            rids = rng.integers(1000, 30000, size=(n_r, S)).astype(np.int32) * (np.arange(S)[None, :] < rl[:, None])
            prov.embed_token_arrays(rids, rl)       # warm: allocator pools for these tile shapes
            t0 = time.perf_counter()
            for _ in range(3):
                prov.embed_token_arrays(rids, rl)
            sorted_s = (time.perf_counter() - t0) / 3
            prov.encoder.close()
            # (c) the whole provider call on text: 1000-character chunks (the reference's splitter setting) ->
            # host WordPiece (libarchi_hip.so, all host cores) -> tiles -> GPU -> float32 rows on the host
            text_rate = tok_rate = None
            try:
                import tempfile
                from archi_amd.ingest import prepare_file
                from tests.synth_text import make_files, make_vocab_file
                with tempfile.TemporaryDirectory() as td:
                    tprov = ArchiHipEmbeddings(name, model_kwargs={"synthetic_seed": 0, "device": f"cuda:{local_rank}",
                                                                   "vocab_file": make_vocab_file(os.path.join(td, "vocab.txt"))},
                                               encode_kwargs={"normalize_embeddings": True})
                    chunks = []
                    for fh, fn, text in make_files(7, 180):
                        chunks += prepare_file(fh, fn, text, "bench")[0]
                    tprov.embed_documents_array(chunks)
                    t0 = time.perf_counter()
                    for _ in range(3):
                        tprov.embed_documents_array(chunks)
                    text_rate = 3 * len(chunks) / (time.perf_counter() - t0)
                    t0 = time.perf_counter()
                    for _ in range(3):
                        tprov.tokenizer.encode_batch_array(chunks, S)
                    tok_rate = 3 * len(chunks) / (time.perf_counter() - t0)
                    tprov.encoder.close()
            except Exception as e:                  # context only
                text_rate = str(e)[:200]
            res["ragged"] = {"lengths": f"uniform in [32,{S}] tokens (mean {float(lens.mean()):.0f})",
                             "padded_tile_chunks_per_s": B / padded_s,
                             "length_sorted_tiles_chunks_per_s": n_r / sorted_s,
                             "text_chunks_per_s": text_rate, "host_tokenizer_chunks_per_s": tok_rate,
                             "host_cores": os.cpu_count(),
                             "note": "per GPU; (b) includes host tiling of the token rows and the D2H of the embeddings; text_chunks_per_s is "
                                     "embed_documents on ~850-character chunks of synthetic text (tokenizer + tiles + GPU + D2H)"}
        except Exception as e:                      # context only
            res["ragged"] = {"error": str(e)[:200]}
    enc.close()
    # second shape of SURVEY 8d: bge-base architecture, [128,512] per rank (CLS pooling); same protocol, fewer steps
    try:
        name2 = "BAAI/bge-base-en"
        vocab2, H2, L2, heads2, I2, max_pos2, pooling2, S2 = MODEL_SHAPES[name2]
        B2 = 128
        weights2 = random_init_weights(vocab2, H2, L2, I2, max_pos2, seed=0)
        enc2 = HipEncoder(vocab2, H2, L2, heads2, I2, max_pos2, weights2, device=local_rank)
        ids2_h = rng.integers(1000, 30000, size=(B2, S2)).astype(np.int32)
        ids2 = torch.from_numpy(ids2_h).cuda()
        mask2 = torch.ones((B2, S2), dtype=torch.int32, device="cuda")
        with GpuTelemetry(local_rank) as tele2:
            el2, steps2, step_ms2, warm2 = timed_loop(lambda: enc2.forward(ids2, mask2, pooling=pooling2), world,
                                                      min_steps=max(30, args.steps))
        cps2 = world * B2 * steps2 / el2
        fl2 = S2 * L2 * (2 * (4 * H2 * H2 + 2 * H2 * I2) + 4 * S2 * H2)
        res["bge_base"] = {"metric": "chunk-embeds/sec (512-token chunks)", "value": cps2, "unit": "chunks/s",
                           "ms_per_step": el2 * 1e3 / steps2, "steps": steps2, "warmup": warm2,
                           "step_ms_hip_events": _ev_summary(step_ms2),
                           "config": {"workload": f"bge-base-en architecture (12 layers, H=768, 12 heads, FFN 3072), random-init, "
                                                  f"{B2} x {S2} synthetic token ids per rank, CLS pooling + L2 normalise"},
                           "roofline": {"bound": "mfma", "achieved": cps2 / world * fl2 / 1e12, "peak": MFMA_BF16_PEAK_TFS,
                                        "unit": "TFLOP/s", "frac": cps2 / world * fl2 / 1e12 / MFMA_BF16_PEAK_TFS,
                                        "algorithmic_flops_per_chunk": fl2}}
        res["bge_base"]["roofline"].update(tele2.summary())
        add_clock_adjusted(res["bge_base"]["roofline"])
        res["bge_base"]["roofline"]["vendor_gemm_tflops"] = vg
        res["bge_base"]["roofline"]["achieved_over_vendor_gemm"] = (res["bge_base"]["roofline"]["achieved"] / vg) if vg else None
        if with_cpu and rank == 0:
            # what the timed batch returned, against the torch-fp32 CPU restatement on the same ids: 3 of the 128 chunks (a chunk's
            # embedding does not depend on its neighbours), so the check goes through the kernels the full batch launches
            from oracle import encoder_oracle as eo
            pick = [0, B2 // 2, B2 - 1]
            got2 = enc2.forward(ids2, mask2, pooling=pooling2).cpu().numpy()[pick]
            w2 = {k: (v if isinstance(v, np.ndarray) else np.asarray(v)) for k, v in weights2.items()}
            ref2 = eo.forward("bge-base", w2, ids2_h[pick], np.ones((len(pick), S2), np.int32), pooling=pooling2)
            cos2 = (got2 * ref2).sum(1) / (np.linalg.norm(got2, axis=1) * np.linalg.norm(ref2, axis=1))
            res["bge_base"]["verified"] = {"chunks": pick, "min_cosine_gpu_vs_cpu": float(cos2.min()),
                                           "max_abs_diff": float(np.abs(got2 - ref2).max()),
                                           "what": "rows of the full 128 x 512 batch against the torch-fp32 CPU restatement"}
            if cos2.min() < 1 - 2e-4:            # (the parity gate proper is tests/: 1 - 1e-4; full-length 512-token rows sit at ~5e-5)
                raise SystemExit(f"bench: bge-base embeddings differ from the CPU restatement (min cosine {cos2.min()})")
        enc2.close()
    except Exception as e:                          # secondary shape: report, never fail the bench
        res["bge_base"] = {"error": str(e)[:200]}
    if rank == 0:
        try:
            res["f32_parity"] = f32_parity_leg(local_rank)
        except Exception as e:                      # secondary leg: report, never fail the bench
            res["f32_parity"] = {"error": str(e)[:200]}
    return res


def f32_parity_leg(local_rank):
    """The float32 parity mode of the encoder (precision="f32": float32 weights, activations and accumulation on
    v_mfma_f32_32x32x2_f32 -- csrc/encoder_f32.hip), the one mode that reproduces the reference's torch-fp32 CPU embedder to
    north_star's 1e-5 from text (embed.end_to_end.f32). Both encoder shapes at the bench's batch sizes, HIP events over a few
    steps, against the 157.3 TFLOP/s float32 matrix roof. Outside the headline metric (that is the bf16 path)."""
    from archi_amd.encoder import MODEL_SHAPES, HipEncoder, random_init_weights
    out = {"what": "encoder forward in float32 throughout (v_mfma_f32_32x32x2_f32), same batches as the bf16 legs; `bf16x3`: the "
                   "split-bf16 parity mode (float32 weights and activations, every GEMM as hi.hi + lo.hi + hi.lo on "
                   "v_mfma_f32_32x32x16_bf16 into one float32 accumulator, attention / LayerNorm / GELU in float32): algorithmic "
                   "flops as the other legs (the three passes are not counted three times), against the bf16 roof",
           "peak_tflops": MFMA_F32_PEAK_TFS}
    for key, name, B, steps, precision in (("minilm", "sentence-transformers/all-MiniLM-L6-v2", 256, 5, "f32"),
                                           ("bge_base", "BAAI/bge-base-en", 128, 3, "f32"),
                                           ("minilm", "sentence-transformers/all-MiniLM-L6-v2", 256, 5, "bf16x3"),
                                           ("bge_base", "BAAI/bge-base-en", 128, 3, "bf16x3")):
        vocab, H, L, heads, I, max_pos, pooling, S = MODEL_SHAPES[name]
        enc = HipEncoder(vocab, H, L, heads, I, max_pos, random_init_weights(vocab, H, L, I, max_pos, seed=0), device=local_rank,
                         precision=precision)
        rng = np.random.default_rng(5)
        ids = torch.from_numpy(rng.integers(1000, 30000, size=(B, S)).astype(np.int32)).cuda()
        mask = torch.ones((B, S), dtype=torch.int32, device="cuda")
        for _ in range(2):
            enc.forward(ids, mask, pooling=pooling)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a, b in ev:
            a.record(); enc.forward(ids, mask, pooling=pooling); b.record()
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        fl = S * L * (2 * (4 * H * H + 2 * H * I) + 4 * S * H)
        tf = B * fl / (ms * 1e-3) / 1e12
        if precision == "f32":
            out[key] = {"batch": f"{B} x {S}", "ms_per_step": ms, "chunks_per_s": B / (ms * 1e-3), "tflops": tf,
                        "frac": tf / MFMA_F32_PEAK_TFS, "steps": steps}
        else:
            out.setdefault("bf16x3", {"peak_tflops": MFMA_BF16_PEAK_TFS})[key] = {
                "batch": f"{B} x {S}", "ms_per_step": ms, "chunks_per_s": B / (ms * 1e-3), "tflops": tf,
                "frac": tf / MFMA_BF16_PEAK_TFS, "speedup_over_f32_mode": out[key]["ms_per_step"] / ms, "steps": steps}
        enc.close()
    return out


def query_latency_leg(local_rank, with_cpu, n_rows=1_000_000, dim=384, n_queries=200, k=4):
    """The reference's real operating point (round-4 review): ONE query per request thread against a table of ~1M chunks --
    embed_query -> one SELECT ... ORDER BY distance LIMIT k -> Documents (postgres_vectorstore.py:227-248,317-332;
    src/interfaces/chat_app/app.py:1554), k = 4 (the retriever default). Here: 200 different text queries through
    ArchiHipVectorStore.similarity_search_with_score on a 1M x 384 float32 collection (random unit vectors, synthetic chunk
    texts, MiniLM-shape encoder with random-init weights and the synthetic WordPiece vocabulary), wall clock per call, p50 / p99,
    and the median of each part measured on its own: host tokeniser, embed_query (tokenise + H2D + forward + D2H), the index
    search through the host-buffer entry point, Document assembly. Beside it the CPU path on the same inputs: the torch-fp32
    encoder oracle on one query + the oracle's sequential scan of a row slice, scaled to the table."""
    import tempfile
    from archi_amd import vectorstore as vs
    from archi_amd.embeddings import ArchiHipEmbeddings
    from tests.synth_text import make_files, make_vocab_file
    name = "sentence-transformers/all-MiniLM-L6-v2"
    with tempfile.TemporaryDirectory() as td:
        prov = ArchiHipEmbeddings(name, model_kwargs={"synthetic_seed": 0, "device": f"cuda:{local_rank}",
                                                      "vocab_file": make_vocab_file(os.path.join(td, "vocab.txt"))},
                                  encode_kwargs={"normalize_embeddings": True})
        store = vs.ArchiHipVectorStore({"hip": {"dtype": "f32", "capacity": n_rows + 1024}}, prov, collection_name="bench_latency")
        t0 = time.perf_counter()
        per, blk = 1000, 50
        g = torch.Generator(device="cuda").manual_seed(99)
        for d0 in range(0, n_rows // per, blk):
            nd = min(blk, n_rows // per - d0)
            v = torch.randn(nd * per, dim, device="cuda", generator=g)
            v = (v / v.norm(dim=1, keepdim=True)).cpu().numpy()
            store.add_texts_batch([([f"chunk {(d0 + j) * per + i} of the synthetic corpus" for i in range(per)],
                                    [{"source": "web", "resource_hash": f"h{d0 + j}", "filename": f"f{d0 + j}.txt"} for _ in range(per)],
                                    d0 + j + 1, v[j * per:(j + 1) * per]) for j in range(nd)])
        build_s = time.perf_counter() - t0
        words = make_files(3, 2, mean_chunks=4.0)[0][2].replace(".", " ").split()
        rng = np.random.default_rng(11)
        queries = [" ".join(rng.choice(words, size=int(rng.integers(5, 13)))) for _ in range(n_queries)]
        for q in queries[:20]:
            store.similarity_search_with_score(q, k=k)
        lat = []
        for q in queries:
            t0 = time.perf_counter()
            res = store.similarity_search_with_score(q, k=k)
            lat.append(time.perf_counter() - t0)
        assert len(res) == k

        def med_ms(fn, items):
            ts = []
            for it in items:
                t0 = time.perf_counter(); fn(it); ts.append(time.perf_counter() - t0)
            return float(np.median(ts) * 1e3)
        S = prov.max_seq_length
        tok_ms = med_ms(lambda q: prov.tokenizer.encode_batch_array([q], S), queries)
        emb_ms = med_ms(lambda q: prov.embed_query(q), queries)
        vecs = [prov.embed_query(q) for q in queries[:50]]
        col = store._collection()
        idx_ms = med_ms(lambda v: col.index.search(np.asarray(v, np.float32)[None], k), vecs)
        byvec_ms = med_ms(lambda v: store.similarity_search_by_vector_with_score(v, k=k), vecs)
        out = {"what": f"similarity_search_with_score(text, k={k}) on a {n_rows} x {dim} float32 collection, {n_queries} different queries, "
                       "one at a time (wall clock per call, Python included)",
               "p50_ms": float(np.percentile(lat, 50) * 1e3), "p99_ms": float(np.percentile(lat, 99) * 1e3),
               "mean_ms": float(np.mean(lat) * 1e3),
               "parts_median_ms": {"tokenise": tok_ms, "embed_query_total": emb_ms, "embed_query_gpu_and_copies": emb_ms - tok_ms,
                                   "index_search_host_buffers": idx_ms, "materialise_documents": max(byvec_ms - idx_ms, 0.0)},
               "build_s": build_s}
        if with_cpu:
            try:
                from archi_amd.encoder import MODEL_SHAPES, random_init_weights
                from oracle import encoder_oracle as eo
                from oracle import knn_oracle as ko
                vocab, H, L, heads, I, max_pos, pooling, _ = MODEL_SHAPES[name]
                w = {kk: np.asarray(vv) for kk, vv in random_init_weights(vocab, H, L, I, max_pos, seed=0).items()}
                ids, lens = prov.tokenizer.encode_batch_array(queries[:4], S)
                Sq = max(32, (int(lens.max()) + 31) // 32 * 32)
                msk = (np.arange(Sq)[None, :] < lens[:, None]).astype(np.int32)
                torch.set_num_threads(min(os.cpu_count() or 1, 64))
                eo.forward("minilm-l6", w, ids[:1, :Sq], msk[:1])
                t0 = time.perf_counter()
                for j in range(4):
                    ref = eo.forward("minilm-l6", w, ids[j:j + 1, :Sq], msk[j:j + 1])
                enc_ms = (time.perf_counter() - t0) / 4 * 1e3
                sl = 100_000
                rows = col.index.fetch(np.arange(sl))
                t0 = time.perf_counter()
                ko.search(rows, ref, k, "cosine")
                scan_ms = (time.perf_counter() - t0) * 1e3 * n_rows / sl
                out["cpu_path"] = {"what": "torch-fp32 encoder oracle on one query (all cores) + the oracle's sequential float32 scan "
                                           f"(one core, what one Postgres backend does) of {sl} rows scaled to {n_rows}",
                                   "embed_query_ms": enc_ms, "scan_ms": scan_ms, "total_ms": enc_ms + scan_ms,
                                   "this_build_p50_over_cpu": (enc_ms + scan_ms) / out["p50_ms"]}
            except Exception as e:                  # context only
                out["cpu_path"] = {"error": str(e)[:200]}
        # the single-launch query forward (csrc/query_forward.hip; opt-in because it is SLOWER on this part): embed_query with it against
        # the default 47 launches, same queries, rows bit-identical -- reported so that the negative result is in the driver's line
        try:
            from archi_amd import _lib as ak_lib
            want = [np.asarray(prov.embed_query(q), np.float32) for q in queries[:20]]
            ak_lib.debug_set("AK_QUERY_FUSED", "2")
            try:
                same = all(np.array_equal(np.asarray(prov.embed_query(q), np.float32), w) for q, w in zip(queries[:20], want))
                fused_ms = med_ms(lambda q: prov.embed_query(q), queries)
            finally:
                ak_lib.debug_set("AK_QUERY_FUSED", None)
            out["single_launch_forward"] = {"what": "embed_query with the whole forward pass as ONE launch confined to one XCD (AK_QUERY_FUSED; opt-in) "
                                                    "against the default multi-launch path above",
                                            "embed_query_total_ms": fused_ms, "default_embed_query_total_ms": emb_ms,
                                            "rows_bit_identical": bool(same), "default": "multi-launch (faster on MI355X)"}
        except Exception as e:                      # context only
            out["single_launch_forward"] = {"error": str(e)[:200]}
        prov.encoder.close()
        vs.reset_collections()
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start N child ranks (one process per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment -- what torch.distributed.run would set), relay rank 0's JSON line and exit
    with the worst child status. This process has not touched the GPU (torch.cuda.device_count() does not initialise it)
    and never does: it only waits. Children are fresh processes, nothing is exec'ed over a process that holds the GPU."""
    import socket
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s); refusing to run fewer ranks than asked", file=sys.stderr)
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0, _ = procs[0].communicate()
    worst = procs[0].returncode
    for pr in procs[1:]:
        rc = pr.wait()
        if rc != 0 and worst == 0:
            worst = rc
    text = out0.decode("utf-8", "replace")
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    for ln in text.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    elif worst == 0:
        worst = 1
    sys.exit(worst if worst >= 0 else 1)


def main():
    args = parse()
    if args.cpu_worker:
        return _cpu_worker(args.cpu_worker)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)              # plain `python bench.py --gpus N`: start the N ranks ourselves
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launcher and flag disagree)")
    # The contract is ONE JSON line on stdout. Libraries write there too (RCCL prints a version banner from C when a communicator
    # is created or torn down, after Python's last print as likely as before it): from here on file descriptor 1 IS stderr, and
    # the line goes to a duplicate of the real stdout, as the last thing this process does.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))      # "nccl" IS RCCL on ROCm

    from archi_amd import _lib
    from archi_amd.index import HipIndex
    from archi_amd.sharded import AbiShardedSearcher, HipLocalSearch, ShardedSearcher, shard_bounds

    _lib.init(local_rank)
    lo, hi = shard_bounds(args.rows, world, rank)
    ix = HipIndex(args.dim, max(hi - lo, 1), dtype=args.dtype, metric="cosine", device=local_rank)
    ix.generate(seed=1234, n=hi - lo, stream=0, row0=lo, normalise=True, id0=lo)

    q_host = gen_queries(args.queries, args.dim, args.dtype)
    q_dev = torch.from_numpy(q_host).cuda()
    local = HipLocalSearch(ix)
    # scan the shard -> ONE all-gather (ids, distances, certificate flags) -> merge; queries some shard could not certify are
    # re-run exactly (none on this corpus). --comm abi: the same sequence issued by the library itself (csrc/shardcomm.hip)
    searcher = AbiShardedSearcher(ix) if args.comm == "abi" else ShardedSearcher(local)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        ids, dd = searcher.search(q_dev, args.k)
    sync_all()
    ix.profile(True)
    searcher.total_open = 0
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    # SURVEY 8d's protocol ends a step with the results on the host: ids + float8 distances of every step are copied D2H inside
    # the timed region (pinned buffers, the search's stream; 164 KB per 1024-query step), behind the step's closing event
    ids_host = torch.empty((args.queries, args.k), dtype=torch.int64).pin_memory()
    dd_host = torch.empty((args.queries, args.k), dtype=torch.float64).pin_memory()
    with GpuTelemetry(local_rank) as tele:
        t0 = time.perf_counter()
        for i in range(args.steps):
            ev[i][0].record()                    # the stream the search is launched on (torch's current stream)
            ids, dd = searcher.search(q_dev, args.k)
            ids_host.copy_(ids, non_blocking=True); dd_host.copy_(dd, non_blocking=True)
            ev[i][1].record()
        sync_all()
        elapsed = time.perf_counter() - t0
        if world == 1 and elapsed < 1.0:         # the timed region is a fraction of a second: keep the same load on, UNTIMED, until
            t_end = time.perf_counter() + 1.0    # the sampler has seen a steady state (these searches are not part of `value`)
            while time.perf_counter() < t_end:
                searcher.search(q_dev, args.k)
            torch.cuda.synchronize()
    step_ms = np.array([a.elapsed_time(b) for a, b in ev]) if args.steps else np.zeros(0)
    scan_ms = ix.profile_read()
    ix.profile(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    cert = int(local.last_cert.sum().item()) if local.last_cert is not None else args.queries - int(searcher.last_open)
    reran = searcher.total_open

    ms_per_step = elapsed * 1e3 / args.steps
    qps = args.queries * args.steps / elapsed
    shard_rows = hi - lo
    plan = ix.scan_plan(args.queries, args.k)
    main_rows = shard_rows - plan["seed_rows"]               # rows covered by the timed (main-pass) launch
    flops = 2.0 * args.queries * main_rows * args.dim        # algorithmic flops of that k_scan launch
    bytes_alg = float(main_rows) * args.dim * 2              # its corpus rows streamed once
    mean_scan_ms = float(scan_ms.mean()) if scan_ms.size else float("nan")
    ridge = MFMA_BF16_PEAK_TFS * 1e12 / (HBM_PEAK_GBS * 1e9)  # flops per byte
    if flops / bytes_alg >= ridge:
        roof = {"bound": "mfma", "achieved": flops / (mean_scan_ms * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFS,
                "unit": "TFLOP/s"}
    else:
        roof = {"bound": "hbm", "achieved": bytes_alg / (mean_scan_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["kernel"] = f"k_scan<{args.dtype}, {plan['cfg_name']}> main pass over {main_rows} of {shard_rows} rows"
    roof["algorithmic_flops_per_launch"] = flops
    roof["algorithmic_bytes_per_launch"] = bytes_alg
    roof["launch_ms"] = mean_scan_ms
    roof["launch_ms_median"] = float(np.median(scan_ms)) if scan_ms.size else None
    roof["launches_timed"] = int(scan_ms.size)
    roof.update(tele.summary())                              # scalars: a slow box shows here, slow code does not
    add_clock_adjusted(roof)
    roof["vendor_gemm_tflops"] = roof["achieved_over_vendor_gemm"] = None
    if rank == 0:
        try:
            ref = vendor_gemm_tflops()
            roof["vendor_gemm_tflops"] = ref                 # torch.matmul bf16 8192^3 (hipBLASLt) on this GPU, same run
            args.vendor_gemm_tflops = ref
            if roof["bound"] == "mfma":
                roof["achieved_over_vendor_gemm"] = roof["achieved"] / ref
            roof["vendor_gemm"] = {"what": "torch.matmul bf16 8192^3 (hipBLASLt) on this GPU, same run", "tflops": ref}
        except Exception as e:                      # context only: never fail the bench for it
            roof["vendor_gemm"] = {"error": str(e)}
    roof["traffic"] = None
    key = f"{args.rows}x{args.dim}_{args.dtype}_q{args.queries}_g{world}"
    roof["traffic_key"] = key                                # scripts/collect_profiles.py files the PMC result under it
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            if key in t:
                roof["traffic"] = t[key]["hbm_bytes_per_launch"]
                roof["traffic_source"] = t[key].get("source")
        except Exception:
            pass

    which = {10_000_000: "configs[2]", 50_000_000: "configs[3]"}.get(args.rows, "custom size")
    out = {
        "metric": f"kNN queries/sec @ top-{args.k}, {args.rows // 1_000_000}M x {args.dim} {args.dtype} corpus",
        "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{which}: {args.rows} x {args.dim} {args.dtype} corpus (unit rows, counter-based "
                               f"Philox generator, seed 1234), cosine top-{args.k}, {args.queries}-query batches, "
                               f"row-sharded over {world} GPU(s); exact results (MFMA candidate scan + re-rank in the "
                               f"reference arithmetic + certificate; uncertified queries re-run exactly)",
                   "rows": args.rows, "dim": args.dim, "queries_per_step": args.queries, "k": args.k,
                   "rows_per_gpu": shard_rows, "scan_plan": plan, "parallelism": f"row-shard x{world} + RCCL all-gather of partial top-k",
                   "comm": "C ABI (ak_index_search_sharded_dev: ncclAllGather issued by libarchi_hip.so)" if args.comm == "abi"
                           else "torch.distributed all_gather_into_tensor (RCCL)"},
        "step_ms_hip_events": {"median": float(np.median(step_ms)) if step_ms.size else None,
                               "min": float(step_ms.min()) if step_ms.size else None,
                               "max": float(step_ms.max()) if step_ms.size else None, "n": int(step_ms.size),
                               "note": "per-step HIP events on the launch stream (rank 0); `value` is wall clock over all steps, max over ranks"},
        "certified_queries_last_step": cert,
        "queries_rerun_exactly_in_timed_steps": reran,
        "roofline": roof,
    }
    ids_h, dd_h = ids.cpu().numpy(), dd.cpu().numpy()
    if not args.no_verify:
        full = ix
        ok = True
        if world > 1 and rank == 0:
            full = HipIndex(args.dim, args.rows, dtype=args.dtype, metric="cosine", device=local_rank)
            full.generate(seed=1234, n=args.rows, stream=0, row0=0, normalise=True, id0=0)
            fi, fd, _ = full.search(q_host, args.k, mode="auto")
            ok = bool(np.array_equal(fi, ids_h) and np.array_equal(fd, dd_h))
            out["sharded_equals_single_index"] = ok
        if rank == 0:
            if not ok:
                raise SystemExit("bench.py verify: sharded result differs from the single-index result")
            out["verified"] = verify_results(args, full, q_host, ids_h, dd_h, args.k)
            if full is not ix:
                full.close()
        if world > 1:
            dist.barrier()
    if rank == 0 and world == 1:
        # PCIe-inclusive rate: the host-buffer entry point -- H2D of the queries, the search, D2H of ids + distances
        try:
            for _ in range(2):
                ix.search(q_host, args.k, mode="auto")
            ts = []
            for _ in range(10):
                t0 = time.perf_counter()
                ix.search(q_host, args.k, mode="auto")
                ts.append(time.perf_counter() - t0)
            med = float(np.median(ts))
            out["pcie_inclusive"] = {"what": "ak_index_search with host buffers (H2D queries + search + D2H results), median of 10 calls",
                                     "ms_per_step": med * 1e3, "queries_per_s": args.queries / med}
        except Exception as e:                      # context only
            out["pcie_inclusive"] = {"error": str(e)[:200]}
    if world > 1 and args.rows * args.dim * 2 <= 64e9:
        # The other way to use N GPUs for a corpus that fits one of them (15 GB here, 288 GB of HBM per GPU): every rank
        # holds the WHOLE corpus and answers its own query batches -- no exchange step at all. Same latency as one GPU,
        # N x the throughput; the row-sharded line above is the configuration SURVEY 8e / BASELINE cfg4 name (it also cuts
        # latency and is the only choice once the corpus outgrows one GPU). Reported beside it, never as `value`.
        try:
            full = HipIndex(args.dim, args.rows, dtype=args.dtype, metric="cosine", device=local_rank)
            full.generate(seed=1234, n=args.rows, stream=0, row0=0, normalise=True, id0=0)
            fl = HipLocalSearch(full)
            q_own = torch.from_numpy(gen_queries(args.queries, args.dim, args.dtype, batch=rank)).cuda()
            for _ in range(args.warmup):
                fl(q_own, args.k, mode="auto")
            sync_all()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fl(q_own, args.k, mode="auto")
            sync_all()
            rel = time.perf_counter() - t0
            tmax = torch.tensor([rel], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            rel = float(tmax.item())
            out["replicated_corpus"] = {"what": f"every rank holds all {args.rows} rows and answers its own {args.queries}-query "
                                                f"batches (query-parallel replicas, no collective)",
                                        "value": world * args.queries * args.steps / rel, "unit": "queries/s",
                                        "ms_per_step": rel * 1e3 / args.steps, "scaling": "weak"}
            full.close()
        except Exception as e:                      # secondary mode: report, never fail the bench
            out["replicated_corpus"] = {"error": str(e)[:200]}
    if rank == 0 and world == 1:
        try:
            v = vendor_knn_qps(args.queries, args.dim, args.k, args.rows)
            out["vendor_stack"] = {"what": "torch bf16 matmul + torch.topk on PyTorch-ROCm, same GPU, 1M-row slice scaled to the "
                                           "corpus (approximate scores, no exact re-rank, slice merge not charged)",
                                   "queries_per_s": v, "this_build_over_vendor_stack": qps / v}
        except Exception as e:                      # context only
            out["vendor_stack"] = {"error": str(e)[:200]}
    if rank == 0 and world == 1 and args.rows == 10_000_000 and not args.no_side_configs:
        try:
            out["hbm_bound_configs"] = hbm_bound_configs(ix, args)
        except Exception as e:                      # context only
            out["hbm_bound_configs"] = {"error": str(e)[:200]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(ix, q_host, args.k, args.rows, args.cpu_seconds)
            out["gpu_over_cpu"] = qps / out["cpu_baseline"]["value"]
        except Exception as e:                      # a reported context number: a host that cannot run it must not fail the GPU measurement
            out["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": None, "kind": "port", "sample": None,
                                   "error": f"{type(e).__name__}: {e}"[:300]}
            out["gpu_over_cpu"] = None
    ix.close()
    ix = None
    if rank == 0 and world == 1 and not args.no_side_configs:
        try:
            out["query_latency"] = query_latency_leg(local_rank, with_cpu=not args.no_cpu_baseline)
        except Exception as e:                      # secondary leg: report, never fail the bench
            out["query_latency"] = {"error": str(e)[:300]}
    if not args.no_embed:
        try:
            out["embed"] = embed_bench(args, world, rank, local_rank, with_cpu=(world == 1 and not args.no_cpu_baseline))
        except Exception as e:                      # the side legs report their failure; the parity guards inside exit (SystemExit) and still do
            if world > 1:
                raise                               # (ranks must not diverge around the leg's barriers)
            out["embed"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if world > 1:
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)


if __name__ == "__main__":
    main()
