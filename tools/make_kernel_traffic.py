"""profiles/kernel_traffic.json (what bench.py reads for `roofline.traffic`) from the round's own PMC passes (VERDICT r5 "next" #7).

    python tools/make_kernel_traffic.py fp32=profiles/r6_pmc_traffic_by_kernel_fp32.json bf16=profiles/r6_pmc_traffic_by_kernel_bf16.json

Input: the per-kernel-name records tools/pmc_bench.sh writes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KB -> B, the gfx950
x2 correction of FETCH_SIZE applied there).  Output: the same figures keyed by bench.py's profiling ids where the kernel template
determines the id; every record names the file it came from."""
import json
import re
import sys


def prof_ids(name):
    """profiling ids (ctl_prof_begin / ctl_prof_begin_raw) a rocprofv3 kernel name stands for"""
    m = re.match(r"conv_igemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false), (true|false), (true|false)>", name)
    if m:
        ks, s, mode, mt, tw, nt, epi, x2, x3, pc = m.groups()
        sfx = (f",e{epi}" if epi != "0" else "") + (",x2" if x2 == "true" else "")
        fam = "conv_igemm" + ("_x3pc" if pc == "true" else "_x3" if x3 == "true" else "")
        return [f"{fam}<ks{ks},s{s},in{mode},mt{mt},tw{tw},nt{nt}{sfx}>"]
    m = re.match(r"conv_wgrad_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false)", name)
    if m:
        ks, s, mode, mt, tw, nt, dy2 = m.groups()
        return [f"conv_wgrad<ks{ks},s{s},in{mode},mt{mt},tw{tw},nt{nt}{',x2' if dy2 == 'true' else ''}>"]
    m = re.match(r"conv_wgrad_x3_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false)", name)
    if m:
        ks, s, mode, mt, nt, dy2 = m.groups()
        return [f"conv_wgrad_x3<ks{ks},s{s},in{mode},mt{mt},tw16,nt{nt}{',x2' if dy2 == 'true' else ''}>"]
    m = re.match(r"conv_wgrad_x3pc16_kernel<(\d+), (\d+), (\d+), (true|false)>", name)
    if m:                                                  # the grouped launches (ctl_conv_wgrad_group); a single-problem launch of this kernel carries the x3 id
        ks, s, mode, dy2 = m.groups()
        return [f"conv_wgrad_x3grp<in{mode}{',x2' if dy2 == 'true' else ''}>"]
    m = re.match(r"conv_igemm_bf16_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false)", name)
    if m and m.group(7) != "0":
        ks, s, mode, mt, tw, nt, fast, xb, x2 = m.groups()
        sfx = (f",e{fast}" if fast != "1" else "") + (",x2" if x2 == "true" else "")
        return [f"conv_igemm_bf16<ks{ks},s{s},in{mode},mt{mt},tw{tw},nt{nt}{sfx}>"]
    m = re.match(r"conv_wgrad_bf16_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false)", name)
    if m:
        ks, s, mode, mt, nt, dy2 = m.groups()
        return [f"conv_wgrad_bf16<ks{ks},s{s},in{mode},mt{mt},tw16,nt{nt}{',x2' if dy2 == 'true' else ''}>"]
    if name.startswith("latent_mask_image_kernel"):
        return ["latent_mask_fused"]
    return []


def main():
    out = {}
    for arg in sys.argv[1:]:
        dt, path = arg.split("=", 1)
        src = json.load(open(path))
        kernels = {}
        for name, rec in src.items():
            for pid in prof_ids(name):
                if pid in kernels:                         # several templates behind one id: keep the one with more launches
                    if kernels[pid]["launches_sampled"] >= rec["launches"]:
                        continue
                kernels[pid] = {"hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "fetch_bytes_per_launch": rec["fetch_bytes_per_launch"],
                                "write_bytes_per_launch": rec["write_bytes_per_launch"], "launches_sampled": rec["launches"],
                                "rocprof_kernel": name[:160], "source": path}
        out[dt] = {"source": f"{path}: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes; KB -> B, FETCH_SIZE x2 on gfx950), "
                             "tools/pmc_bench.sh on this round's build", "kernels": kernels}
    json.dump(out, open("profiles/kernel_traffic.json", "w"), indent=1)
    for dt, v in out.items():
        print(dt, len(v["kernels"]), "profiling ids")


if __name__ == "__main__":
    main()
