"""What the amdgpu hwmon files say while the scan / the encoder runs (bench.py's GpuTelemetry reads them): a 10 ms series of
freq1_input, freq2_input, power1_input around a 2 s load.   python3 scripts/gpu_probe_telemetry.py"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from archi_amd.index import HipIndex
hw = [h for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")) if os.path.exists(h + "/freq1_input")]
print("hwmon:", hw)
for h in hw:
    for f in ("freq1_label", "freq2_label", "power1_label", "power1_cap", "name"):
        try: print(" ", f, open(f"{h}/{f}").read().strip())
        except OSError: pass
import torch as _t
pr = _t.cuda.get_device_properties(0); pci = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
print("device 0 pci", pci, [(x, os.path.basename(os.path.realpath(x.split("/hwmon/")[0]))) for x in hw])
h = next((x for x in hw if os.path.basename(os.path.realpath(x.split("/hwmon/")[0])).lower().startswith(pci)), hw[0])
print("reading", h)
rd = lambda f: float(open(f"{h}/{f}").read())
series, stop = [], threading.Event()
def run():
    t0 = time.perf_counter()
    while not stop.is_set():
        try: series.append((time.perf_counter() - t0, rd("freq1_input") / 1e6, rd("freq2_input") / 1e6, rd("power1_input") / 1e6))
        except Exception as e: series.append((time.perf_counter() - t0, str(e)))
        time.sleep(0.01)
ix = HipIndex(768, 4_000_000, dtype="bf16", metric="cosine", device=0); ix.generate(seed=1, n=4_000_000, normalise=True)
q = np.random.default_rng(0).standard_normal((1024, 768)).astype(np.float32)
th = threading.Thread(target=run); th.start()
time.sleep(0.3); t_load = time.perf_counter()
while time.perf_counter() - t_load < 2.0: ix.search(q, 10)
time.sleep(0.3); stop.set(); th.join()
for s in series[::4]: print("  t=%.2f sclk %.0f MHz mclk %.0f MHz power %.0f W" % s if len(s) == 4 else s)
