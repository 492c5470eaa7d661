"""bench.py's GpuTelemetry must read the hwmon files of the BOUND device: a box's sysfs lists every GPU of the host (round 5: the
first visits reported another card's 158 MHz / 269 W as the clock and power under load). The card is matched by PCI address;
no match means no telemetry, never a neighbour's."""
import importlib.util
import os
import time


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _card(root, name, pci, mhz, watts, cap):
    dev = root / "devices" / pci
    hw = dev / "hwmon" / f"hwmon{name[-1]}"
    hw.mkdir(parents=True)
    (hw / "freq1_input").write_text(str(int(mhz * 1e6)))
    (hw / "power1_input").write_text(str(int(watts * 1e6)))
    (hw / "power1_cap").write_text(str(int(cap * 1e6)))
    card = root / "drm" / name
    card.mkdir(parents=True)
    os.symlink(dev, card / "device")


def test_telemetry_reads_the_card_with_the_devices_pci_address(tmp_path):
    b = _bench()
    _card(tmp_path, "card0", "0000:75:00.0", 158, 269, 1400)
    _card(tmp_path, "card7", "0000:a7:00.0", 1790, 1230, 1400)
    with b.GpuTelemetry(0, period_s=0.005, sysfs_root=str(tmp_path / "drm"), pci="0000:a7:00") as t:
        time.sleep(0.05)
    s = t.summary()
    assert s["sclk_mhz_under_load"] == 1790 and s["power_w_under_load"] == 1230 and s["power_cap_w"] == 1400
    assert s["telemetry_source"].endswith("hwmon7") and s["telemetry_pci"] == "0000:a7:00" and s["telemetry_samples"] >= 2


def test_no_matching_card_means_no_telemetry(tmp_path):
    b = _bench()
    _card(tmp_path, "card0", "0000:75:00.0", 158, 269, 1400)
    with b.GpuTelemetry(0, period_s=0.005, sysfs_root=str(tmp_path / "drm"), pci="0000:01:00") as t:
        time.sleep(0.02)
    s = t.summary()
    assert s["sclk_mhz_under_load"] is None and s["power_w_under_load"] is None and s["telemetry_source"] is None


def test_a_hung_cpu_worker_ends_its_probe_not_the_bench():
    """cpu_baseline's layout sweep: a worker that never reports (seen on one box in round 5: TimeoutExpired took the whole bench
    down) makes its layout NaN within the deadline, every child of the probe is killed and reaped, and a healthy probe still sums."""
    import json
    import sys
    b = _bench()
    ok = [sys.executable, "-c", "import json; print(json.dumps({'qrows_per_s': 2.5}))"]
    hang = [sys.executable, "-c", "import time; time.sleep(600)"]
    assert b.run_cpu_workers([ok, ok], dict(os.environ), 30.0) == 5.0
    t0 = time.perf_counter()
    r = b.run_cpu_workers([ok, hang, hang], dict(os.environ), 2.0)
    assert r != r and time.perf_counter() - t0 < 20.0
    bad = [sys.executable, "-c", "print('not json')"]
    r = b.run_cpu_workers([ok, bad], dict(os.environ), 30.0)
    assert r != r
