"""dev: do the shards' uploads overlap?  `python tools/pin_overlap.py run [rows]` pins a three-column host table with the operator layer's current shard
setting (RFX_SHARDS / RFX_DEVICES) and prints the wall time; run it under `rocprofv3 --memory-copy-trace --output-format csv -d DIR -- python
tools/pin_overlap.py run` and then `python tools/pin_overlap.py report DIR` summarises the trace: host-to-device copies per destination stream / agent, and
for how much of the upload's span 1, 2, 3, 4 ... copies were in flight at once (the reference maps every column file where it lies, core/io.c:1310-1364;
here every shard's row range goes through its own stream -- on an 8-device node its own copy engine and PCIe link -- at the same time)."""
import csv, glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(rows):
    import numpy as np
    from oracle import rfo
    from rayforce_amd import hostobj as H
    ops = H.lib()
    ops.rfx_host_bind()
    host = {"k": rfo.gen_i64(rows, 4, 1_000_000), "a": rfo.gen_i64(rows, 2, 1_000_000), "v": rfo.gen_f64(rows, 5)}
    tab = H.table(host)
    warm = H.vector(np.arange(1 << 23, dtype=np.int64))  # contexts, staging buffers and the worker set come up here, not inside the timed pin
    ops.rfx_host_drop(ops.rfx_pin(warm))
    ops.rfx_host_drop(ops.rfx_unpin(warm))
    ops.rfx_host_drop(warm)
    t0 = time.perf_counter()
    p = ops.rfx_pin(tab)
    dt = time.perf_counter() - t0
    assert p and not H.is_error(p), H.error_text(p)
    print(f"rfx_pin of 3 x {rows} rows ({3 * rows * 8 / 1e9:.1f} GB) over {ops.rfx_ops_shards()} shard(s): {dt * 1e3:.0f} ms = {3 * rows * 8 / dt / 1e9:.1f} GB/s", flush=True)
    d = H.select_dict({"where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v")}, tab)
    r = ops.rfx_select(d)
    want = rfo.select({"from": host, "where": ("<", "a", 100_000), "by": "k", "s": ("sum", "v")})
    got = H.table_to_numpy(r)
    assert np.array_equal(got["k"], want["k"]) and np.allclose(got["s"], want["s"], rtol=1e-9, atol=0)
    print("answer unchanged (group keys in order, sums within 1e-9 of the oracle)")


def report(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    h2d = [r for r in rows if "HOST_TO_DEVICE" in (r.get("Direction") or r.get("Kind") or "").upper() or "H2D" in (r.get("Direction") or "").upper()]
    if not h2d:
        print("no host-to-device copies in the trace; columns:", list(rows[0]) if rows else "(empty)")
        return
    ev = []
    for r in h2d:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e - s < 200_000:  # (only the 32 MB staging chunks: >= 0.2 ms each)
            continue
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, hist = 0, None, {}
    for t, dlt in ev:
        if last is not None and depth > 0:
            hist[depth] = hist.get(depth, 0) + (t - last)
        depth += dlt
        last = t
    tot = sum(hist.values())
    print(f"{len(ev) // 2} staging-chunk copies; time with k copies in flight (of {tot / 1e6:.1f} ms with any):")
    for k in sorted(hist):
        print(f"  {k}: {hist[k] / 1e6:8.2f} ms  {100.0 * hist[k] / tot:5.1f} %")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(float(sys.argv[2])) if len(sys.argv) > 2 else 200_000_000)
    else:
        report(sys.argv[2])
