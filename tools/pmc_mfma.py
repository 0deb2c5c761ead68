#!/usr/bin/env python3
"""MFMA-pipe occupancy per kernel from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES,
SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY): tools/pmc_mfma.py <dir> [<dir> ...].
busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): GRBM_GUI_ACTIVE is summed over the 8 XCDs
(it gives 2.5 GHz x the kernel's duration when divided by 8), the busy cycles over all SIMDs (= MFMA count x cycles each)."""
import csv, glob, sys
from collections import defaultdict
for root in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
        per = defaultdict(lambda: defaultdict(float))
        for row in csv.DictReader(open(f)):
            per[(row["Kernel_Name"], row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
        for (k, _), cs in per.items():
            for c, v in cs.items():
                acc[k][c].append(v)
    print("==", root)
    rows = []
    for k, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        if m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
            continue
        gui = m["GRBM_GUI_ACTIVE"] / 8.0
        rows.append((m["SQ_VALU_MFMA_BUSY_CYCLES"] * len(cs["GRBM_GUI_ACTIVE"]), k, m, gui))
    for _, k, m, gui in sorted(rows, reverse=True)[:8]:
        wc = m["SQ_WAVE_CYCLES"]
        print("%-96s" % k[:96])
        print("    launches %4d   GUI_ACTIVE/8 %9.0f cycles   MFMA busy %12.0f   busy fraction %.3f   wave cycles: issue %.2f, "
              "issue-stall %.2f, parked %.2f" % (len(acc[k]["GRBM_GUI_ACTIVE"]), gui, m["SQ_VALU_MFMA_BUSY_CYCLES"],
                                               m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0), m["SQ_ACTIVE_INST_ANY"] / wc,
                                               m["SQ_WAIT_INST_ANY"] / wc, m["SQ_WAIT_ANY"] / wc))
