"""dev: every kernel of librfx.so with its register count and scratch bytes per lane (from the code objects' metadata): `python tools/kernel_resources.py [min scratch]`.
A kernel whose register TILE ends up in scratch (a per-lane select chain folded into one indexed load, DESIGN section 3 "compiler traps") shows here as scratch = tile bytes + a few."""
import os, re, struct, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(root, "rayforce_amd", "librfx.so")
LLVM = "/opt/rocm/lib/llvm/bin"
minscratch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
with tempfile.TemporaryDirectory() as td:
    fb = os.path.join(td, "fb.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", so, os.path.join(td, "x.so")], check=True)
    data = open(fb, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    rows = []
    pos = 0
    while True:
        at = data.find(magic, pos)
        if at < 0:
            break
        n, = struct.unpack_from("<Q", data, at + 24)
        p = at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                co = os.path.join(td, "k.co")
                open(co, "wb").write(data[at + off:at + off + size])
                notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
                name = None
                cur = {}
                for line in notes.splitlines():
                    m = re.match(r"\s+-?\s*\.(name|private_segment_fixed_size|vgpr_count|sgpr_spill_count|vgpr_spill_count):\s+(\S+)", line)
                    if not m:
                        continue
                    cur[m.group(1)] = m.group(2)
                    if len(cur) == 5:
                        rows.append(cur)
                        cur = {}
        pos = at + 24
    rows = [r for r in rows if int(r.get("private_segment_fixed_size", 0)) >= minscratch]
    rows.sort(key=lambda r: -int(r["private_segment_fixed_size"]))
    for r in rows:
        demangled = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        print(f"scratch {int(r['private_segment_fixed_size']):5d}  vgpr {int(r['vgpr_count']):3d}  vgpr spills {int(r['vgpr_spill_count']):3d}  sgpr spills {int(r['sgpr_spill_count']):3d}  {demangled[:150]}")
    print(f"{len(rows)} kernels with >= {minscratch} bytes of scratch per lane")
