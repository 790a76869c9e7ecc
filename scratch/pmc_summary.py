import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (d, r["Dispatch_Id"])
            if key not in seen and d == sys.argv[1]:
                seen.add(key); calls[k] += 1
names = sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CYCLES", 0))
cols = ["SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_VALU_MFMA_BUSY_CYCLES","SQ_INSTS_VALU","SQ_INSTS_MFMA","SQ_INSTS_LDS","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_INSTS_VMEM","SQ_WAIT_INST_LDS","FETCH_SIZE","WRITE_SIZE","TCC_HIT_sum","TCC_MISS_sum","GRBM_GUI_ACTIVE"] + [c for c in sorted({c for k in agg for c in agg[k]}) if c.startswith(("TA_", "TCP_", "TCC_REQ", "TCC_READ", "TCC_EA", "TCC_TAG", "TCC_BUSY", "TD_"))]
print("per-dispatch averages (counter sum / dispatches); FETCH/WRITE_SIZE in KB (FETCH_SIZE under-reads wide streams 2x on gfx950)")
print("%-52s %5s " % ("kernel", "n") + " ".join("%12s" % c.replace("SQ_", "")[:12] for c in cols))
for k in names[:40]:
    n = max(calls[k], 1)
    print("%-52s %5d " % (k[:52], n) + " ".join("%12.4g" % (agg[k].get(c, 0) / n) for c in cols))
