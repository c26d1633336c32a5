"""gpurun_out/r3p (tools/profile_r03.sh) -> profiles/r03_*: the rocprofv3 kernel-stats CSVs as they are, and one JSON per
workload with the PMC counters of the dominant kernel(s) and what follows from them.  Units as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: WRITE_SIZE counts KiB, FETCH_SIZE counts KiB and reads half."""
import csv
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "gpurun_out" / "r3p"
DST = ROOT / "profiles"
PEAK = 39.3e12


def counters(d):
    tot = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for f in (SRC / d).rglob("*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            tot[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[row["Kernel_Name"]].add(row["Dispatch_Id"])
    return tot, {k: len(v) for k, v in disp.items()}


def stats(d):
    f = next((SRC / d).rglob("*kernel_stats.csv"))
    return {r["Name"]: r for r in csv.DictReader(open(f))}, f


def pair(tag, kernel_prefix, m, n, K, ops, label=None, args="", note=""):
    label = label or f"nw{tag}"
    st, f = stats(f"stats_{tag}")
    shutil.copy(f, DST / f"r03_{label}_kernel_stats.csv")
    name = next(k for k in st if k.startswith(kernel_prefix))
    avg_ms = float(st[name]["AverageNs"]) / 1e6
    c = defaultdict(float)
    for p in ("write", "fetch", "sq"):
        t, nd = counters(f"pmc_{p}_{tag}")
        assert nd[name] == 1, nd
        c.update(t[name])
    S = -(-m // (64 * K))
    cells = m * n
    wbytes, fbytes = c["WRITE_SIZE"] * 1024, c["FETCH_SIZE"] * 1024 * 2
    algo = cells * 0.25 + 2 * 8.0 * (n + 64) * S
    out = {
        "command": "tools/profile_r03.sh: one rocprofv3 pass each for --pmc WRITE_SIZE | --pmc FETCH_SIZE | --pmc SQ_* GRBM_GUI_ACTIVE on "
                   f"`{note}python3 bench.py --no-cpu --steps 1 --warmup 0 --no-batch --no-small{args}`; kernel-trace stats from a run of its own",
        "workload": f"{label} {m}x{n}, persistent strip pipeline, K={K}, {S} strips, one launch; {avg_ms:.2f} ms (kernel-trace average, r03_{label}_kernel_stats.csv)",
        "kernel": name,
        "counters": dict(sorted(c.items())),
        "derived": {
            "WRITE_SIZE_bytes": wbytes,
            "FETCH_SIZE_bytes_x2_gfx950_correction": fbytes,
            "hbm_traffic_bytes_per_launch": wbytes + fbytes,
            "algorithmic_bytes_per_launch": algo,
            "algorithmic_note": "2-bit trace codes (0.25 B/cell) + the bottom row of every strip written once and read once as 8-byte records",
            "traffic_over_algorithmic": (wbytes + fbytes) / algo,
            "valu_instr_per_cell_lane": c["SQ_INSTS_VALU"] * 64 / cells,
            "wave_cycles_issuing_valu": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] if c["SQ_WAVE_CYCLES"] else None,
            "wave_cycles_waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c["SQ_WAVE_CYCLES"] else None,
            "wave_cycles_waiting_for_an_instruction_to_issue": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"] if c["SQ_WAVE_CYCLES"] else None,
            "waves": c["SQ_WAVES"],
            "roofline_frac_at_kernel_trace_time": cells * ops / (avg_ms / 1e3) / PEAK,
        },
    }
    (DST / f"r03_{label}_pmc.json").write_text(json.dumps(out, indent=1))
    print(tag, name, f"{avg_ms:.2f} ms", {k: round(v, 4) for k, v in out["derived"].items() if isinstance(v, float) and v < 100})


def batch(pairs, calls):
    st, f = stats("stats_batch")
    shutil.copy(f, DST / "r03_batch_kernel_stats.csv")
    c = defaultdict(float)
    for p in ("write", "fetch", "sq"):
        t, nd = counters(f"pmc_{p}_batch")
        for k in t:
            if "sw_batch_kernel" in k:
                for name, v in t[k].items():
                    c[name] += v
    wbytes, fbytes = c["WRITE_SIZE"] * 1024 / calls, c["FETCH_SIZE"] * 1024 * 2 / calls
    from sequencealigner_amd.synth import protein_pairs

    _pool, _ox, lx, _oy, ly = protein_pairs(pairs, seed=42)
    algo = float(lx.sum() + ly.sum()) + 44.0 * pairs
    kern_ms = sum(float(r["TotalDurationNs"]) / float(r["Calls"]) for k, r in st.items() if "sw_batch_kernel" in k) / 1e6
    out = {"batch": {
        "command": "tools/profile_r03.sh: rocprofv3 --pmc passes on `python3 bench.py --no-cpu --steps 1 --warmup 0 --workload batch` "
                   f"({calls} calls of sa_sw_batch: pool resident, then pool on the host); counters summed over the 11 sw_batch_kernel<K, false> instances, per call",
        "pairs": pairs, "calls": calls, "kernel_ms_per_call_kernel_trace": kern_ms,
        "counters_all_calls": dict(sorted(c.items())),
        "WRITE_SIZE_bytes_per_call": wbytes, "FETCH_SIZE_bytes_x2_gfx950_correction_per_call": fbytes,
        "hbm_traffic_bytes_per_call": wbytes + fbytes,
        "algorithmic_bytes_per_call": algo,
        "algorithmic_note": "every sequence byte once + 32 B of descriptor in and 12 B of result out per pair",
        "traffic_over_algorithmic": (wbytes + fbytes) / algo,
    }}
    (DST / "r03_batch_pmc.json").write_text(json.dumps(out, indent=1))
    print("batch", {k: v for k, v in out["batch"].items() if isinstance(v, float)})


def lcs_bits(m, n):
    st, f = stats("stats_lcs1m")
    shutil.copy(f, DST / "r03_lcs1m_kernel_stats.csv")
    name = next(k for k in st if "lcs_bits_persistent" in k)
    avg_ms = float(st[name]["AverageNs"]) / 1e6
    c = defaultdict(float)
    for p in ("write", "fetch", "sq"):
        t, nd = counters(f"pmc_{p}_lcs1m")
        assert nd[name] == 1, nd
        c.update(t[name])
    S, steps = -(-m // 2048), ((n + 126 + 63) // 64) * 64
    wbytes, fbytes = c["WRITE_SIZE"] * 1024, c["FETCH_SIZE"] * 1024 * 2
    algo = float(S) * steps * 256
    out = {
        "command": "tools/profile_r03.sh: one rocprofv3 pass each for --pmc WRITE_SIZE | --pmc FETCH_SIZE | --pmc SQ_* GRBM_GUI_ACTIVE on "
                   "`python3 bench.py --no-cpu --steps 1 --warmup 0 --no-batch --kind lcs`; kernel-trace stats from a run of its own",
        "workload": f"LCS {m}x{n} as bit vectors (sa_lcsbits.hip), {S} strips of 2048 rows, {steps} steps each, one launch; {avg_ms:.2f} ms "
                    "(kernel-trace average, r03_lcs1m_kernel_stats.csv)",
        "kernel": name,
        "counters": dict(sorted(c.items())),
        "derived": {
            "WRITE_SIZE_bytes": wbytes,
            "FETCH_SIZE_bytes_x2_gfx950_correction": fbytes,
            "hbm_traffic_bytes_per_launch": wbytes + fbytes,
            "algorithmic_bytes_per_launch": algo,
            "algorithmic_note": "the traceback record: one dword per lane per step = 1 bit per cell (+ the lane skew and the padding of the last strip)",
            "traffic_over_algorithmic": (wbytes + fbytes) / algo,
            "hbm_GBps_at_kernel_trace_time": algo / (avg_ms / 1e3) / 1e9,
            "valu_instr_per_column_step": c["SQ_INSTS_VALU"] / (S * steps),
            "salu_instr_per_column_step": c["SQ_INSTS_SALU"] / (S * steps),
            "wave_cycles_issuing_valu": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] if c["SQ_WAVE_CYCLES"] else None,
            "wave_cycles_waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c["SQ_WAVE_CYCLES"] else None,
            "wave_cycles_per_column_step": c["SQ_WAVE_CYCLES"] / (S * steps),
        },
    }
    (DST / "r03_lcs1m_pmc.json").write_text(json.dumps(out, indent=1))
    print("lcs1m", name, f"{avg_ms:.2f} ms", {k: round(v, 4) for k, v in out["derived"].items() if isinstance(v, float) and v < 1e6})


sys.path.insert(0, str(ROOT))
lcs_bits(1_000_000, 999_747)
pair("1m", "void sa::fill_persistent<0, 8, 0>", 1_000_000, 999_747, 8, 12)
pair("100k", "void sa::fill_persistent<0, 2, 0>", 100_000, 100_085, 2, 12, args=" --size 100000")
pair("sw1m", "void sa::fill_persistent<1, 8, 0>", 1_000_000, 999_747, 8, 14, label="sw1m", args=" --kind sw")
# the second wave per SIMD: the same pair on fill_group with two workgroups per CU, and on one
for wps in (2, 1):
    try:
        pair(f"1m_group_wps{wps}", "void sa::fill_group<0, 8, 0>", 1_000_000, 999_747, 8, 12, label=f"nw1m_group_wps{wps}", note=f"SA_GROUP=1 SA_WPS={wps} ")
    except StopIteration:  # (a stats-only refresh, `profile_r03.sh s`, does not run the fill_group pair)
        print(f"1m_group_wps{wps}: not in this collection, profiles/ keeps the earlier files")
for tag, dst in (("100k_unrel", "nw100k_unrelated"), ("2m", "nw2m_checkpointed"), ("lcs2m", "lcs2m_checkpointed"), ("1m_unrel", "nw1m_unrelated")):
    st, f = stats(f"stats_{tag}")
    shutil.copy(f, DST / f"r03_{dst}_kernel_stats.csv")
batch(1_000_000, 2)
