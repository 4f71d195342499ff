"""Post-processing of tools/pmc_step.sh: per rocprofv3 kernel name the mean counter values per launch, derived fractions, and the in-process
profiling id (bench.py's `roofline.kernel` vocabulary) where the template arguments determine it."""
import collections, csv, glob, json, re, sys

out, tag, bench_args = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{out}/p1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)


def prof_id(name):
    m = re.match(r"conv_igemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false)(?:, (true|false))?", name)
    if m:
        ks, s, mode, mt, tw, nt, epi, x2, x3 = m.groups()
        sfx = (f",e{epi}" if epi != "0" else "") + (",x2" if x2 == "true" else "")
        return f"conv_igemm{'_x3' if x3 == 'true' else ''}<ks{ks},s{s},in{mode},mt{mt},tw{tw},nt{nt}{sfx}>"
    m = re.match(r"conv_wgrad_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false)", name)
    if m:
        ks, s, mode, mt, tw, nt, dy2 = m.groups()
        return f"conv_wgrad<ks{ks},s{s},in{mode},mt{mt},tw{tw},nt{nt}{',x2' if dy2 == 'true' else ''}>"
    m = re.match(r"conv_wgrad_x3_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false)", name)
    if m:
        ks, s, mode, mt, nt, dy2 = m.groups()
        return f"conv_wgrad_x3<ks{ks},s{s},in{mode},mt{mt},tw16,nt{nt}{',x2' if dy2 == 'true' else ''}>"
    m = re.match(r"conv_igemm_bf16_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false)", name)
    if m and m.group(7) != "0":
        ks, s, mode, mt, tw, nt, fast, xb, x2 = m.groups()
        sfx = (f",e{fast}" if fast != "1" else "") + (",x2" if x2 == "true" else "")
        return f"conv_igemm_bf16<ks{ks},s{s},in{mode},mt{mt},tw{tw},nt{nt}{sfx}>"
    m = re.match(r"conv_wgrad_bf16_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false)", name)
    if m:
        ks, s, mode, mt, nt, dy2 = m.groups()
        return f"conv_wgrad_bf16<ks{ks},s{s},in{mode},mt{mt},tw16,nt{nt}{',x2' if dy2 == 'true' else ''}>"
    return None


res = {}
for k, cs in acc.items():
    mean = {c: sum(v) / len(v) for c, v in cs.items()}
    n = max(len(v) for v in cs.values())
    rec = {"launches": n, "prof_id": prof_id(k), "avg_us_under_pmc": sum(dur[k]) / len(dur[k]) if dur.get(k) else None, "counters_per_launch": mean}
    if "FETCH_SIZE" in mean:      # KB -> B; gfx950 counts half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM / rocprofv3 section)
        rec["fetch_bytes_per_launch"] = mean["FETCH_SIZE"] * 1024 * 2
        rec["write_bytes_per_launch"] = mean.get("WRITE_SIZE", 0.0) * 1024
        rec["hbm_bytes_per_launch"] = rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"]
    if mean.get("SQ_BUSY_CYCLES"):
        # SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4): / 32 = the launch's busy cycles (checked against the trace duration x the
        # shader clock: 67.7 k cycles for a 37.4 us launch = 1.81 GHz); SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (= N_mfma x 16 for 16x16x32 bf16)
        rec["launch_cycles"] = mean["SQ_BUSY_CYCLES"] / 32
        rec["mfma_busy_frac"] = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (rec["launch_cycles"] * 1024)
    if mean.get("SQ_LDS_IDX_ACTIVE"):
        rec["lds_bank_conflict_frac"] = mean.get("SQ_LDS_BANK_CONFLICT", 0.0) / mean["SQ_LDS_IDX_ACTIVE"]
    if mean.get("SQ_WAVE_CYCLES"):
        for c, key in (("SQ_WAIT_INST_ANY", "wait_inst_frac"), ("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_ACTIVE_INST_VALU", "valu_active_frac")):
            rec[key] = mean.get(c, 0.0) / mean["SQ_WAVE_CYCLES"]
    res[k] = rec
meta = {"method": "rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub-records --mode eager "
                  + bench_args + " ; five passes (FETCH_SIZE | WRITE_SIZE | MFMA busy | LDS + waits | instruction mix); FETCH_SIZE x 1024 x 2, WRITE_SIZE x 1024; "
                  "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 shader engines x 1024 SIMDs); wait / active fractions over SQ_WAVE_CYCLES", "tag": tag}
json.dump({"meta": meta, "kernels": res}, open(f"{out}/counters_by_kernel.json", "w"), indent=1)
rows = sorted(res.items(), key=lambda kv: -(kv[1]["launches"] * (kv[1]["avg_us_under_pmc"] or 0)))
with open(f"{out}/summary.txt", "w") as f:
    f.write(f"{'kernel':84s} {'n':>5s} {'us':>7s} {'HBM MB':>8s} {'mfma':>6s} {'ldsconf':>7s} {'w_inst':>6s} {'w_any':>6s} {'valu':>6s}\n")
    for k, r in rows[:40]:
        f.write(f"{(r['prof_id'] or k)[:84]:84s} {r['launches']:5d} {(r['avg_us_under_pmc'] or 0):7.1f} {r.get('hbm_bytes_per_launch', 0) / 1e6:8.1f} "
                f"{r.get('mfma_busy_frac', 0):6.3f} {r.get('lds_bank_conflict_frac', 0):7.3f} {r.get('wait_inst_frac', 0):6.3f} {r.get('wait_any_frac', 0):6.3f} {r.get('valu_active_frac', 0):6.3f}\n")
print(open(f"{out}/summary.txt").read())
