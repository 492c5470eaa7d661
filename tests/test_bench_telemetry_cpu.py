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
