"""Shader clock and socket power WHILE the dominant kernels run (back "power-limited" with a trace, or retract it).
A sampler child — started before this process touches the GPU, reads only sysfs (hwmon freq1_input / power1_average|input,
pp_dpm_sclk) or, failing that, polls rocm-smi — logs (time, sclk, power) every few ms; the parent then runs each workload in a loop for
about two seconds with wall-clock marks, and the samples inside each window are summarised (median / min / max clock, mean power, achieved
rate).  → gpurun_out/r04_clock_power_trace.json
usage: python tools/clock_power_trace.py"""
import glob, json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def _read(p):
    try:
        with open(p) as f:
            return f.read()
    except Exception:
        return None


def sampler(path):
    """every card of the host (the parent finds out which one it was given only after initialising the GPU)"""
    cards = {}
    for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        if not re.fullmatch(r"card\d+", os.path.basename(c)):
            continue
        pci = os.path.basename(os.path.realpath(os.path.join(c, "device")))
        src = {}
        for h in sorted(glob.glob(os.path.join(c, "device/hwmon/hwmon*"))):
            for k in ("freq1_input", "power1_average", "power1_input"):
                if _read(os.path.join(h, k)) is not None:
                    src.setdefault(k, os.path.join(h, k))
        if src:
            cards[pci] = src
    with open(path, "w") as f:
        f.write("# cards %s\n" % json.dumps(cards)); f.flush()
        while True:
            rec = {"t": time.time()}
            for pci, src in cards.items():
                for k, p in src.items():
                    v = _read(p)
                    if v and v.strip():
                        rec[pci + "/" + k] = int(v.strip())
            f.write(json.dumps(rec) + "\n"); f.flush()
            time.sleep(0.004)


if len(sys.argv) > 2 and sys.argv[1] == "--sampler":
    sampler(sys.argv[2]); sys.exit(0)

os.makedirs(OUT, exist_ok=True)
LOG = os.path.join(OUT, "clock_power_samples.log")
child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--sampler", LOG])      # before any HIP call in this process
try:
    import math
    import torch
    sys.path.insert(0, ROOT)
    from svpc_amd import ops as O
    from svpc_amd.ops_common import SeqInfo
    O.set_precision("bf16x3")
    DEV = "cuda:0"
    torch.zeros(1, device=DEV); torch.cuda.synchronize()
    pr = torch.cuda.get_device_properties(0)
    PCI = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    print("device", pr.name, "at", PCI, flush=True)
    windows = []

    def loop(name, fn, work, unit, seconds=2.0):
        """runs fn in batches of 20 launches until `seconds` have passed; work = flops or bytes per launch"""
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        time.sleep(0.3)                       # back to idle between windows
        t0 = time.time(); n = 0
        per = []
        while time.time() - t0 < seconds:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            per.append((time.time() - t0, e0.elapsed_time(e1) * 50)); n += 20
        t1 = time.time()
        windows.append(dict(name=name, t0=t0, t1=t1, launches=n, us_first=per[0][1], us_last=per[-1][1],
                            us_curve=[(round(a, 3), round(b, 2)) for a, b in per[:: max(1, len(per) // 24)]], work=work, unit=unit))
        print("%-44s first %.1f us  last %.1f us" % (name, per[0][1], per[-1][1]), flush=True)

    def gemm(M, N, K):
        x = O.to_split(torch.randn(M, K, device=DEV))
        w = torch.randn(N, K, device=DEV) / math.sqrt(K)
        b = torch.randn(N, device=DEV) * 0.1
        w16 = O._transient_split(w)
        return lambda: O.linear(x, w, b, w16=w16)

    time.sleep(1.0)
    windows.append(dict(name="idle", t0=time.time() - 0.8, t1=time.time(), launches=0, work=0, unit=""))
    with torch.no_grad():
        loop("gemm_p8x3 19200x2304x768 (Q|K|V)", gemm(19200, 2304, 768), 6.0 * 19200 * 2304 * 768, "issued MFMA FLOP")
        loop("gemm_p8x3 19200x768x768", gemm(19200, 768, 768), 6.0 * 19200 * 768 * 768, "issued MFMA FLOP")
        loop("gemm_p8x3 19200x768x3072 (long K)", gemm(19200, 768, 3072), 6.0 * 19200 * 768 * 3072, "issued MFMA FLOP")
        loop("gemm_p8x3 19200x768x64 (epilogue only)", gemm(19200, 768, 64), 6.0 * 19200 * 768 * 64, "issued MFMA FLOP")
        # an HBM-bound row kernel on the same stream: LayerNorm forward on split rows (reads 4 B + writes 4 B per element)
        xs = O.to_split(torch.randn(19200, 768, device=DEV)); g = torch.ones(768, device=DEV); bt = torch.zeros(768, device=DEV)
        loop("ln_fwd split rows 19200x768", lambda: O.layernorm(xs, g, bt, 1e-12, out_bf16=True), 19200 * 768 * 8.0, "bytes")
        # clip-encoder attention forward (192 sequences x 100 rows x 12 heads)
        qkv = O.to_split(torch.randn(19200, 2304, device=DEV))
        seq = SeqInfo.uniform(192, 100, 100, DEV)
        loop("attn_pipe_fwd x3 192x100x12x64", lambda: O.attention(qkv, qkv, (0, 768, 1536), 768, 12, seq), 19200 * 768 * 4 * 4.0, "bytes")
finally:
    child.terminate(); child.wait()

samples = []
header = None
for line in open(LOG):
    if line.startswith("#"):
        header = header or line.strip(); continue
    try:
        samples.append(json.loads(line))
    except Exception:
        pass
res = {"sampler": header, "device_pci": PCI, "n_samples": len(samples), "windows": []}
samples = [dict(t=s["t"], **{k.split("/", 1)[1]: v for k, v in s.items() if k.startswith(PCI + "/")}) for s in samples]
others = sorted({k.split("/")[0] for s0 in [json.loads(l) for l in open(LOG) if l.startswith("{")][:1] for k in s0 if "/" in k})
res["cards_seen"] = others


def stat(v):
    v = sorted(v)
    return None if not v else dict(median=v[len(v) // 2], min=v[0], max=v[-1], n=len(v))


for w in windows:
    inside = [s for s in samples if w["t0"] + 0.05 <= s["t"] <= w["t1"]]
    late = [s for s in samples if w["t0"] + 0.5 * (w["t1"] - w["t0"]) <= s["t"] <= w["t1"]]
    r = dict(name=w["name"], seconds=round(w["t1"] - w["t0"], 3), launches=w["launches"])
    for key, scale, label in (("freq1_input", 1e-6, "sclk_mhz"), ("power1_average", 1e-6, "power_w"), ("power1_input", 1e-6, "power_w_input")):
        v = [s[key] * scale for s in inside if key in s]
        if v:
            r[label] = stat(v); r[label + "_second_half"] = stat([s[key] * scale for s in late if key in s])
    if w["launches"]:
        r.update(us_first_batch=round(w["us_first"], 2), us_last_batch=round(w["us_last"], 2), us_curve=w["us_curve"], work_per_launch=w["work"],
                 unit=w["unit"], rate_last=w["work"] / (w["us_last"] * 1e-6))
    res["windows"].append(r)
    print(json.dumps({k: v for k, v in r.items() if k != "us_curve"}))
with open(os.path.join(OUT, "r04_clock_power_trace.json"), "w") as f:
    json.dump(res, f, indent=1)
