#!/usr/bin/env python3
"""Per-dispatch timing of ONE denoising step in graph replay, from a `rocprofv3 --kernel-trace` CSV of bench.py:
the trace is cut at the sampler's step kernel, steps with the modal launch count are averaged position by position.
    python tools/step_sequence.py <..._kernel_trace.csv> [--all]
Prints the class totals (and with --all every dispatch in order), i.e. the in-graph budget without per-launch event overhead."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
cut = [i for i, n in enumerate(names) if "ddim_step_kernel" in n or "resshift_step_kernel" in n or "ddpm_step_kernel" in n]
lens = collections.Counter(cut[i + 1] - cut[i] for i in range(len(cut) - 1))
per = lens.most_common(1)[0][0]
good = [(cut[i], cut[i + 1]) for i in range(len(cut) - 1) if cut[i + 1] - cut[i] == per]
acc, gap = [0.0] * per, [0.0] * per
for a, _ in good:
    for k in range(per):
        r = rows[a + 1 + k]
        acc[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        gap[k] += int(r["Start_Timestamp"]) - int(rows[a + k]["End_Timestamp"])
n = len(good)


def short(s):
    s = re.sub(r"^void ", "", s)
    s = re.sub(r"mrisr::", "", s)
    s = re.sub(r"\(.*", "", s)
    m = re.match(r"_ZN5mrisr(\d+)(\w+)", s)
    if m:
        s = m.group(2)[:int(m.group(1))]
    return s[:64]


a0 = good[0][0]
seq = [(short(names[a0 + 1 + k]), acc[k] / n / 1e3, gap[k] / n / 1e3) for k in range(per)]
print(f"{n} steps of {per} launches: busy {sum(d for _, d, _ in seq):.1f} us, gaps {sum(g for _, _, g in seq):.1f} us")
cls = collections.defaultdict(lambda: [0, 0.0])
for nm, d, _ in seq:
    cls[nm][0] += 1
    cls[nm][1] += d
for nm, (c, d) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
    print(f"{nm:66s} {c:4d} {d:9.1f} us {d / c:8.2f} us avg")
if "--all" in sys.argv:
    for k, (nm, d, g) in enumerate(seq):
        print(k, nm, f"{d:.1f}", f"gap {g:.1f}")
