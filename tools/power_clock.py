"""Socket power and shader clock while ONE kernel runs back to back for a few seconds (are the first-layer
GEMMs bound by cycles or by power?).  A sampler thread reads the card's hwmon / rocm-smi files while the main thread keeps
the stream full; nothing here changes a clock or a limit (ordinary user).

    python tools/power_clock.py [seconds per kernel]
"""
import glob, json, os, subprocess, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd import datagen
from cuda_gcn_amd.ops import Device, _ck


def find_sensors():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name, key in (("power1_average", "power_uW"), ("power1_input", "power_uW"), ("freq1_input", "sclk_Hz"), ("power1_cap", "cap_uW")):
            p = os.path.join(hw, name)
            if os.path.exists(p) and key not in out:
                out[key] = p
    return out


def read(p):
    try:
        return float(open(p).read().strip())
    except Exception:
        return None


def smi_sample():
    try:
        o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        return json.loads(o)
    except Exception as e:
        return {"error": repr(e)}


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    sensors = find_sensors()
    print("sensors:", sensors, flush=True)
    print("rocm-smi idle:", json.dumps(smi_sample())[:600], flush=True)
    ds = datagen.make_dataset("reddit-syn")
    N, F, h = ds["num_nodes"], ds["input_dim"], 128
    dev = Device(0); lib = dev.lib
    rng = np.random.default_rng(0)
    f = dev.feat(ds["f_indptr"], ds["f_indices"], ds["f_val"], F)
    g = dev.graph(ds["g_indptr"], ds["g_indices"], row_group=ds["label"])
    w1 = dev.buf(rng.standard_normal((F, h)).astype(np.float32)); h0 = dev.buf(rng.standard_normal((N, h)).astype(np.float32))
    h1 = dev.buf((N, h)); dw = dev.buf((F, h)); ep = dev.buf(np.zeros(1, np.uint32))
    kernels = {
        "forward GEMM (persistent, no dropout)": lambda: _ck(lib, lib.gcnhip_spmm_fwd(dev.ctx, f.h, f.values_ptr, w1.ptr, h, h0.ptr, h, h, 0.0, 1, ep.ptr, 0, None), "f"),
        "forward GEMM (persistent, dropout 0.5)": lambda: _ck(lib, lib.gcnhip_spmm_fwd(dev.ctx, f.h, f.values_ptr, w1.ptr, h, h0.ptr, h, h, 0.5, 1, ep.ptr, 0, None), "f"),
        "backward GEMM (tiles, dropout 0.5)": lambda: _ck(lib, lib.gcnhip_spmm_bwd(dev.ctx, f.h, f.values_ptr, h0.ptr, h, dw.ptr, h, h, 0.5, 1, ep.ptr, 0, None), "b"),
        "GraphSum d=128": lambda: _ck(lib, lib.gcnhip_graphsum(dev.ctx, g.h, h0.ptr, h, h1.ptr, h, h), "g"),
    }
    for name, fn in kernels.items():
        samples, stop = [], threading.Event()

        def sampler():                                        # rocm-smi sees the one visible card; ~0.2 s per call
            while not stop.is_set():
                d = smi_sample().get("card0", {})
                try:
                    samples.append((float(d["Current Socket Graphics Package Power (W)"]), float(d["sclk clock speed:"].strip("()Mhz"))))
                except Exception:
                    pass
        for _ in range(5):
            fn()
        dev.sync()
        th = threading.Thread(target=sampler); th.start()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < secs:
            for _ in range(200):
                fn()
            dev.sync(); n += 200
        dt = time.perf_counter() - t0
        stop.set(); th.join()
        tail = samples[len(samples) // 3:] or samples           # after the first third: settled
        pw = sum(x[0] for x in tail) / max(len(tail), 1)
        ck = sum(x[1] for x in tail) / max(len(tail), 1)
        print(f"{name}: {1e3 * dt / n:.3f} ms per launch over {dt:.1f} s; socket power {pw:.0f} W (cap {read(sensors.get('cap_uW', '')) or 0:.0f} uW); "
              f"sclk {ck:.0f} MHz; {len(tail)} samples, sclk range {min(x[1] for x in tail):.0f}-{max(x[1] for x in tail):.0f}", flush=True)


if __name__ == "__main__":
    main()
