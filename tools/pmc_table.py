#!/usr/bin/env python
"""Compact per-kernel table from the pmc_sq*.txt files of tools/pmc_sq.sh: python tools/pmc_table.py gpurun_out [substring ...]"""
import collections, glob, re, sys
root = sys.argv[1]
subs = sys.argv[2:] or ["chain_kernel", "vlad_accum", "knn_quad", "three_nn_grid", "afa_cluster"]
vals = collections.defaultdict(dict)
for f in sorted(glob.glob(root + "/pmc_sq*.txt")):
    for line in open(f):
        m = re.match(r"(.*?)\s+((?:SQ|GRBM|TCC|TCP)_\w+)\s+n=\s*\d+ avg=([0-9.]+) min=([0-9.]+)", line)
        if m:
            vals[m.group(1).strip()][m.group(2)] = float(m.group(3))
for k, v in vals.items():
    if not any(s in k for s in subs):
        continue
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    gui = v.get("GRBM_GUI_ACTIVE", 0) / 8 or 1
    print(k[:100])
    print("   kernel cycles %.0f  waves %.0f  wave_cycles/wave %.0f (quad-cycles x4 = %.0f cycles)" % (gui, v.get("SQ_WAVES", 0), wc / max(v.get("SQ_WAVES", 1), 1), 4 * wc / max(v.get("SQ_WAVES", 1), 1)))
    print("   of wave cycles: WAIT_ANY %.2f  WAIT_INST_ANY %.2f  ACTIVE_INST_ANY %.2f | VALU %.2f LDS %.2f VMEM %.2f SCA %.2f MISC %.2f  WAIT_INST_LDS %.2f" % tuple(
        v.get(c, 0) / wc for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_LDS")))
    print("   MFMA busy %.3f of SIMD-cycles | insts: VALU %.0f MFMA-mops %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f SALU %.0f | LDS bank conflict cycles %.0f of LDS active %.0f" % (
        v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024), v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0), v.get("SQ_INSTS_LDS", 0), v.get("SQ_INSTS_VMEM_RD", 0),
        v.get("SQ_INSTS_VMEM_WR", 0), v.get("SQ_INSTS_SALU", 0), v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_LDS_IDX_ACTIVE", 0)))
