#!/usr/bin/env python
"""MFMA utilisation per kernel from one rocprofv3 --pmc pass holding SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES-free: python tools/pmc_mfma_util.py x_results.db
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); also SQ_INSTS_VALU_MFMA_MOPS_F32 / F16 per launch where collected."""
import collections, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
v = collections.defaultdict(dict)
for name, cn, n, avg, mn in c.execute("select kernel_name, counter_name, count(*), avg(value), min(value) from counters_collection group by kernel_name, counter_name"):
    v[name][cn] = (avg, mn, n)
print("MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); busy cycles averaged, kernel cycles the minimum over the launches of tools/pmc_target.py")
rows = []
for k, d in v.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["SQ_VALU_MFMA_BUSY_CYCLES"][0] > 0:
        cyc = d["GRBM_GUI_ACTIVE"][1] / 8.0
        rows.append((d["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (cyc * 1024.0), k, cyc, d))
for u, k, cyc, d in sorted(rows, reverse=True):
    extra = "  ".join(f"{cn.replace('SQ_', '')}={d[cn][0]:.0f}" for cn in ("SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT") if cn in d)
    print(f"{u * 100:5.1f}%  {k[:100]:100s} kernel cycles {cyc:9.0f}  {extra}")
