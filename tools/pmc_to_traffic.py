"""Derive profiles/hbm_traffic.json (read by bench.py for `roofline.traffic`) and the compact
profiles/*_pmc_fetch_write.csv from two rocprofv3 counter passes of bench.py:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir>/fetch -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dir>/write -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline
    python tools/pmc_to_traffic.py <dir> profiles r01

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both
counters are in KiB; FETCH_SIZE under-reports by 2x on gfx950.  The factor is cross-checked on
lbm::collide_kernel, whose read traffic is known exactly (36 B per cell + one 8-byte mask word per
64 cells).
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def per_kernel(path):
    d = collections.defaultdict(list)
    rows = []
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        rows.append(r)
    return {k: sum(v) / len(v) for k, v in d.items()}, rows


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)  # a directory may hold several passes
    fetch, frows = per_kernel(newest(os.path.join(src, "fetch", "*", "*_counter_collection.csv")))
    write, wrows = per_kernel(newest(os.path.join(src, "write", "*", "*_counter_collection.csv")))
    keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
            "VGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    csv_name = f"{tag}_bench_pmc_fetch_write.csv"
    with open(os.path.join(dst, csv_name), "w", newline="") as f:
        w = csv.DictWriter(f, keep, quoting=csv.QUOTE_NONNUMERIC)
        w.writeheader()
        for r in frows + wrows:
            w.writerow({k: r[k] for k in keep})

    def find(pat):
        return [k for k in fetch if re.search(pat, k)]

    def total(k):
        return 2 * fetch[k] * 1024 + write[k] * 1024

    step = find(r"lbm::step_kernel<")[0]
    coll = find(r"lbm::collide_kernel<")[0]
    cells = 16384 * 16384
    known_read_kib = (cells * 36 + cells // 64 * 8) / 1024
    out = {
        "bytes_per_launch": total(step),
        "read_bytes_per_launch": 2 * fetch[step] * 1024,
        "write_bytes_per_launch": write[step] * 1024,
        "fetch_size_kib_raw": fetch[step],
        "write_size_kib_raw": write[step],
        "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported; both in KiB",
        "calibration": {
            "kernel": "lbm::collide_kernel (reads exactly 36 B/cell + one 8 B mask word per 64 cells, "
                      "same dword-per-lane access)",
            "known_read_kib": known_read_kib,
            "fetch_size_kib_raw": fetch[coll],
            "ratio": known_read_kib / fetch[coll],
        },
        "workload": "16384x16384 fp32 periodic, T=1, " + step.split("(")[0],
        "temporal_blocking": {},
        "source": f"profiles/{csv_name} (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes "
                  f"of bench.py --steps 6 --warmup 1; derived by tools/pmc_to_traffic.py)",
    }
    for k in sorted(find(r"lbm::step_tb_kernel<")):
        m = re.search(r"step_tb_kernel<([^>]*)>", k)
        T = int(m.group(1).split(",")[1])
        out["temporal_blocking"][m.group(1)] = {
            "bytes_per_launch": total(k), "bytes_per_cell_update": total(k) / cells / T,
            "fetch_size_kib_raw": fetch[k], "write_size_kib_raw": write[k]}
    with open(os.path.join(dst, "hbm_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
