import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref, rfo
n = 300_007
cols = {"k": rfo.gen_i64(n, 4, 5000), "a": rfo.gen_i64(n, 2, 1_000_000), "v": rfo.gen_f64(n, 5)}
for threads in (None, 8, 64):
    s = ref.Session()
    s.table("t", cols)
    qs = ["{s: (sum a) c: (count a) from: t where: (< a 100000)}",
          "{f: (sum v) x: (avg v) mn: (min v) mx: (max v) from: t where: (and (< a 500000) (> v 0.25) (!= k 7))}",
          "{s: (sum v) c: (count a) m: (max a) from: t by: k}",
          "{s: (sum v) from: t where: (> v 0.5) by: k}",
          "{from: t where: (< a 1000)}"]
    for i, q in enumerate(qs):
        s.eval(f'(println "start {i}")')
        s.eval(f"(set g{i} (select {q}))")
        s.eval(f'(println "done {i}")')
    open("/tmp/dd.rfl", "w").write("\n".join(s.lines) + '\n(println "end")\n')
    cmd = [ref.BIN, "-f", "/tmp/dd.rfl"] + (["-c", str(threads)] if threads else [])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=60, stdin=subprocess.DEVNULL)
    print("threads", threads, "rc", p.returncode, p.stdout[-300:].replace("\n", " | "), p.stderr[-300:])
