"""Opt-in diagnostics of bench.py (--diagnostics): the shader clock of the GPU sampled on a side thread during the timed
region, temperatures / power after it, and the workgroup-to-XCD map of a launch.  Moved out of bench.py in round 4
(round-3 verdict: no side thread in the measuring process by default).  They were written for the run-to-run spread of
profiles/r03_NOTES.txt 21, which they do not explain."""
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class SclkSampler:
    """Shader clock of THIS GPU while the timed region runs (sysfs pp_dpm_sclk of the card with the device's PCI
    address, one 30-byte read every 10 ms on a side thread), temperatures and the other clocks once after it.  The boxes
    are GPUs of shared 8-GPU nodes; under this load the clock sits anywhere between ~1.95 and ~2.25 GHz.  Recorded
    because about one call in three reads 10 % slower on the same build (profiles/r03_NOTES.txt 21) - these numbers
    turned out NOT to tell the two kinds of call apart, which is worth knowing too."""

    def __init__(self, dev):
        import glob
        self.path, self.seen, self._stop, self._thread = None, [], None, None
        try:
            pr = torch.cuda.get_device_properties(dev)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
            for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
                if want in os.path.realpath(os.path.dirname(f)):
                    self.path = f
        except Exception:                                  # noqa: BLE001 - a diagnostic, never a reason to fail
            self.path = None

    def _read(self):
        for line in open(self.path).read().splitlines():
            if line.rstrip().endswith("*"):
                return int(line.split(":")[1].lower().split("mhz")[0])
        return None

    def _hwmon(self):
        """socket power (W) and the temperatures the card's hwmon node reports (deg C), read once after the region"""
        import glob
        out = {}
        base = os.path.dirname(self.path)
        for f in glob.glob(os.path.join(base, "hwmon", "hwmon*", "power1_average")):
            out["power_w"] = round(int(open(f).read()) / 1e6, 1)
        for f in sorted(glob.glob(os.path.join(base, "hwmon", "hwmon*", "temp*_input"))):
            lab = f.replace("_input", "_label")
            name = open(lab).read().strip() if os.path.exists(lab) else os.path.basename(f)[:5]
            out["temp_c_" + name] = round(int(open(f).read()) / 1e3, 1)
        for clk in ("mclk", "fclk", "socclk"):
            try:
                for line in open(os.path.join(base, "pp_dpm_" + clk)).read().splitlines():
                    if line.rstrip().endswith("*"):
                        out[clk + "_mhz"] = int(line.split(":")[1].lower().split("mhz")[0])
            except Exception:                              # noqa: BLE001
                pass
        return out

    def start(self):
        if self.path is None:
            return
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                try:
                    v = self._read()
                    if v:
                        self.seen.append(v)
                except Exception:                          # noqa: BLE001
                    pass
                self._stop.wait(0.01)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is None:
            return None
        self._stop.set()
        self._thread.join()
        v = sorted(self.seen)
        if not v:
            return None
        try:
            extra = self._hwmon()
        except Exception:                                  # noqa: BLE001
            extra = {}
        return {"sclk_mhz_median": v[len(v) // 2], "sclk_mhz_min": v[0], "sclk_mhz_max": v[-1], "samples": len(v), **extra,
                "source": "pp_dpm_sclk / hwmon of this GPU, sclk every 10 ms of the timed region; diagnostics for the "
                          "run-to-run spread of profiles/r03_NOTES.txt 21 (which they do not explain)"}


def xcd_map_probe():
    """On which XCD did workgroup b of a 256 x 512-thread launch on THIS stream land (HW_REG_XCC_ID)?  The field kernels'
    schedules take b % 8 (for locality only).  A diagnostic from tools/_probe/libxcdmap.so (tools/micro/xcd_map_probe.hip,
    built here on first use); None when that fails."""
    import ctypes
    import shutil
    import subprocess
    path = os.path.join(ROOT, "tools", "_probe", "libxcdmap.so")
    src = os.path.join(ROOT, "tools", "micro", "xcd_map_probe.hip")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(path) and os.path.exists(src) and os.path.exists(hipcc):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        subprocess.call([hipcc, "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", path])
    if not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
        lib.xcd_map.restype = ctypes.c_int
        lib.xcd_map.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int32),
                                ctypes.c_void_p]
        blocks = torch.cuda.get_device_properties(0).multi_processor_count
        out = (ctypes.c_int32 * blocks)()
        st = torch.cuda.current_stream().cuda_stream
        if lib.xcd_map(blocks, 512, 44 * 1024, 2000, out, ctypes.c_void_p(st)) != 0:
            return None
        ids = [v & 15 for v in out]
        rot = [(ids[b] - b) % 8 for b in range(blocks)]
        hist = [ids.count(x) for x in range(8)]
        return {"xcd_of_block_is_block_mod_8_up_to_rotation": len(set(rot)) == 1, "rotation": rot[0] if len(set(rot)) == 1 else None,
                "workgroups_per_xcd": hist, "first_16_blocks": ids[:16]}
    except Exception as e:                                 # noqa: BLE001 - a diagnostic
        return {"error": f"{type(e).__name__}: {e}"[:120]}
