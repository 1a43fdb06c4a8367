"""Flag kernels whose global loads are serialised by the compiler (load → s_waitcnt vmcnt(0) → load → …): every such pair is a memory round
trip of its own.  usage: python tools/isa_audit.py svpc_amd/csrc/*.hip   (cross-compiles each file to gfx950 assembly with hipcc -S)"""
import re, subprocess, sys, collections
for src in sys.argv[1:]:
    asm = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", "-"],
                         capture_output=True, text=True).stdout
    name, seq = None, []
    kernels = collections.OrderedDict()
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1); seq = kernels.setdefault(name, [])
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith("s_endpgm"):
            name = None
        elif t.startswith(("global_load_dword", "buffer_load_dword")) and "lds" not in t:
            seq.append("L")
        elif t.startswith(("global_store", "buffer_store")):
            seq.append("S")
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            seq.append("W")
    for k, seq in kernels.items():
        s = "".join(seq)
        loads = s.count("L")
        if loads < 4 and s.count("S") < 4:
            continue
        lone = len(re.findall(r"(?<![L])L{1,2}W", "W" + s))     # one or two loads directly followed by a full wait
        rmw = len(re.findall(r"LWS", s))
        sws = len(re.findall(r"SW(?=S)", s))                    # a store, a full drain, the next store
        trips = len(re.findall(r"L+[SL]*W", s))                 # runs of loads each closed by a full drain = memory round trips in the text
        if lone >= int(__import__("os").environ.get("AUDIT_MIN", "4")) or rmw >= int(__import__("os").environ.get("AUDIT_MIN", "4")) or sws >= int(__import__("os").environ.get("AUDIT_MIN", "4")) or trips >= int(__import__("os").environ.get("AUDIT_TRIPS", "6")):
            dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:110]
            print(f"{src.split('/')[-1]:22s} loads {loads:3d}  lone load+wait {lone:3d}  load-wait-store {rmw:3d}  store-wait-store {sws:3d}  load-runs {trips:3d}  {dem}")
