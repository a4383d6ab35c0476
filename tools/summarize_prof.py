"""Summarise a tools/profile.sh output directory: per-kernel time stats and per-dispatch PMC averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


for f in find("trace/**/*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 12:
                print(", ".join(row))
for f in find("trace/**/*kernel_trace.csv"):
    durs = defaultdict(list)
    meta = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            durs[name].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            meta[name] = {k: row.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size")}
    print("== kernel trace:", os.path.relpath(f, out))
    # bench.py's own launches of the rollout kernel, in order: placement candidates (each on its own tensor, the first with a
    # warm-up), then the pre-roll, the W warm-up and the K timed passes on the chosen tensor.  The whole-run average mixes the
    # tensors; the trailing launches are the ones bench.py times.
    seq = []
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "k_rollout" in row["Kernel_Name"]:
                seq.append((int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3, row["Kernel_Name"]))
    seq.sort()
    if seq:
        # the headline's kernel = the rollout symbol launched most often; bench.py's trajectory_full_pass_check adds, after the clock
        # has stopped, one more pass of it and one pass of a twin through the OTHER 2D kernel: both are left out of the tail
        names = [n for _, _, n in seq]
        main = max(set(names), key=names.count)
        twin = [i for i, n in enumerate(names) if n != main]
        own = [d for _, d, n in seq if n == main]
        print("rollout launches in order (us): " + " ".join("%.0f%s" % (d, "" if n == main else "*") for _, d, n in seq) + ("   (*: the check's twin, another kernel)" if twin else ""))
        if twin and twin[-1] == len(seq) - 1 and len(own) > 1:
            own = own[:-1]                                        # the check's own pass of the headline kernel
        if len(own) >= 6:
            tail = own[-6:]
            print("last 6 launches before the check (1 warm-up + 5 timed passes of profile.sh's bench.py --steps 5 --warmup 1, on the chosen tensor): avg %.1f us min %.1f max %.1f"
                  % (sum(tail) / len(tail), min(tail), max(tail)))
    for name, d in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        short = name[:90]
        print("%-90s n=%d avg=%.1f us min=%.1f max=%.1f total=%.3f ms %s" % (short, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, sum(d) / 1e6, meta[name]))
for f in find("pmc_*/**/*counter_collection.csv") + find("p[0-9]*/**/*counter_collection.csv"):
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("== pmc:", os.path.relpath(f, out))
    for name, cs in acc.items():
        if "rollout" not in name and "k_" not in name:
            continue
        for c, v in cs.items():
            print("%-60s %-24s n=%d avg=%.6g" % (name[:60], c, len(v), sum(v) / len(v)))

# HBM traffic per launch of the rollout kernel, as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE come from
# separate --pmc passes, are in KiB, and FETCH_SIZE under-reports wide reads by 2x on gfx950.
import json

vals = defaultdict(dict)                                     # kernel family -> counter -> per-launch average
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in find("pmc_%s/**/*counter_collection.csv" % c):
        per = defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != c:
                    continue
                for fam in ("k_rollout", "k_transition", "k_step"):
                    if fam in row["Kernel_Name"]:
                        nm = row["Kernel_Name"]
                        dim = "K3D" if ("K3D" in nm or "rollout3d" in nm or "step3d" in nm or "transition3d" in nm) else (
                            "K2D" if ("K2D" in nm or "rollout2d" in nm or "step2d" in nm or "transition2d" in nm) else "K1D")
                        per[fam + ":" + dim].append(float(row["Counter_Value"]))
        for fam, v in per.items():
            vals[fam][c] = sum(v) / len(v)
for fam, v in sorted(vals.items()):
    if len(v) == 2:
        t = dict(kernel=fam, fetch_kib=v["FETCH_SIZE"], write_kib=v["WRITE_SIZE"],
                 hbm_bytes_per_launch=(2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0,
                 note="2*FETCH_SIZE + WRITE_SIZE (KiB) per launch; separate --pmc passes; gfx950 read correction x2")
        if fam.startswith("k_rollout"):
            # which kernel source the counters belong to: bench.py drops `roofline.traffic` when the library has changed since
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from snac_amd import _lib
            t["source_sha16"] = _lib.kernel_source_sha16()
            with open(os.path.join(out, "traffic.json"), "w") as fh:
                json.dump(t, fh, indent=1)
        print("== traffic:", t)
