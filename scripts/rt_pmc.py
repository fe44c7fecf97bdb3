#!/usr/bin/env python3
"""More hardware counters of the bench kernel than the bench line carries: where the cycles go that the vector ALUs do not use.
bench.py's own --pmc-child launches under rocprofv3 --pmc, one pass per group; averages per launch of the bench kernel."""
import collections, csv, glob, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
groups = [["GRBM_GUI_ACTIVE", "TA_BUSY_avr", "TA_BUSY_max", "MemUnitStalled", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"],
          ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM"],
          ["SQ_INST_LEVEL_VMEM", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_LATENCY_sum"],
          ["SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_SALU", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS"],
          ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES"],
          ["TA_TA_BUSY_sum", "TD_TD_BUSY_sum", "TCP_GATE_EN1_sum", "TCP_TAGRAM0_REQ_sum", "TA_FLAT_READ_WAVEFRONTS_sum", "TCP_TOTAL_READ_sum"]]
acc = collections.defaultdict(lambda: [0.0, 0])
extra = sys.argv[1:]
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    for gi, g in enumerate(groups):
        out = os.path.join(td, "p%d" % gi)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + g + ["--output-format", "csv", "-d", out, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child"] + extra
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        except subprocess.TimeoutExpired:
            print(json.dumps({"group": g, "error": "timeout"})); continue
        n = 0
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if bench.BENCH_KERNEL in row["Kernel_Name"]:
                    a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1; n += 1
        if n == 0: print(json.dumps({"group": g, "error": "no rows"}))
print(json.dumps({k: round(v[0] / v[1], 1) for k, v in sorted(acc.items())}, indent=1))
