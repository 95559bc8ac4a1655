#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units).
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half of a wide coalesced read stream -> x2.
Writes profiles/pmc_traffic.json (read by bench.py for roofline.traffic) and prints a markdown table."""
import csv, glob, json, os, subprocess, sys, collections
sys.path.insert(0, os.getcwd())
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = max(glob.glob(f"gpurun_out/pmc_{c}/*/*_counter_collection.csv"), key=os.path.getmtime)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[name][0] += float(r["Counter_Value"]); acc[name][1] += 1
    out[c] = acc
rows = []
for name in out["FETCH_SIZE"]:
    fs, n = out["FETCH_SIZE"][name]; ws, n2 = out["WRITE_SIZE"].get(name, [0.0, 1])
    rd = 2 * fs * 1024 / n; wr = ws * 1024 / max(n2, 1)
    rows.append((name, n, rd, wr))
rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
print("| kernel | launches | read B/launch (2xFETCH_SIZE) | written B/launch | total GB/launch |\n|---|---|---|---|---|")
for name, n, rd, wr in rows[:16]:
    print(f"| {name[:60]} | {n} | {rd:.3e} | {wr:.3e} | {(rd+wr)/1e9:.3f} |")
rp = [r for r in rows if r[0].startswith("row_pass_kernel")]
if rp:
    import bench
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
        dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "crosslayer_transcoder_amd/csrc"], capture_output=True, text=True).stdout.strip())
    except OSError:
        commit, dirty = "unknown", False
    json.dump({"row_pass_bytes_per_launch": rp[0][2] + rp[0][3], "read": rp[0][2], "written": rp[0][3], "launches": rp[0][1],
               "commit": commit + ("+uncommitted kernel changes" if dirty else ""), "kernel_source_digest": bench.kernel_source_digest(),
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB*1024, FETCH_SIZE x2 (gfx950 wide-stream correction)"},
              open("profiles/pmc_traffic.json", "w"))
