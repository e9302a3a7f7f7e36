"""Summarises scripts/stage_budget.sh: per-wave instruction counts of the bounce-2 launch of k_bounce<false> for each
stage limit, and the per-stage differences."""
import collections, csv, glob, sys
root = sys.argv[1]
rows = {}
for stage in (1, 2, 3, 0):
    files = glob.glob(f"{root}/s{stage}/**/*_counter_collection.csv", recursive=True)
    if not files:
        continue
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(files[0])):
        if "k_bounce<false" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)
    picked = [per[i] for k, i in enumerate(ids) if k % 4 == 1]   # bounce 2 of every chunk
    acc = collections.defaultdict(float)
    for p in picked:
        for c, v in p.items():
            acc[c] += v
    waves = acc["SQ_WAVES"]
    rows[stage] = {c: v / waves for c, v in acc.items() if c != "SQ_WAVES"}
    rows[stage]["launches"] = len(picked)
    rows[stage]["waves_per_launch"] = waves / max(1, len(picked))
names = {1: "load + cull loop", 2: "+ ray->local, sphere, box stage", 3: "+ triangle stage (whole cast)", 0: "+ shade, Philox, compaction, store (whole kernel)"}
cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]
print("per wave of the bounce-2 launch of k_bounce<false> (cumulative)")
print(f"{'stage':52s}" + "".join(f"{c[9:]:>10s}" for c in cols))
prev = None
for stage in (1, 2, 3, 0):
    if stage not in rows:
        continue
    print(f"{names[stage]:52s}" + "".join(f"{rows[stage].get(c, 0):10.1f}" for c in cols))
print("\nper stage (differences)")
for stage in (1, 2, 3, 0):
    if stage not in rows:
        continue
    base = prev or {c: 0.0 for c in cols}
    print(f"{names[stage]:52s}" + "".join(f"{rows[stage].get(c, 0) - base.get(c, 0):10.1f}" for c in cols))
    prev = rows[stage]
print("\nlaunches averaged:", {s: r["launches"] for s, r in rows.items()}, "waves per launch:", {s: round(r["waves_per_launch"]) for s, r in rows.items()})
