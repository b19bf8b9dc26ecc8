#!/usr/bin/env python
"""profiles/traffic_<config>.json from the PMC passes of scripts/gpu_pmc.sh: measured HBM-side bytes per LAUNCH of the
dominant kernel groups (`sgm`: the sgm_pass_kernel launches of one step; `cbca`: the kernels of one iteration over one volume), collected and
corrected as MI355X_MICROARCH.md prescribes: separate --pmc passes, FETCH_SIZE/WRITE_SIZE in KiB, FETCH_SIZE x2 on
gfx950 (128-byte requests tallied as 64 B), WRITE_SIZE at face value.
  python scripts/make_traffic_json.py gpurun_out/<tag>/pmc_<config> <config>"""
import collections
import csv
import glob
import json
import os
import sys


def main(d, config):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "p*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfile = os.path.join(root, "mc-cnn_amd", "BUILD_COMMIT")   # written in the build container before the snapshot travels (the GPU box has no .git)
    out = {"_note": "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024, mean over dispatches; source: " + d,
           "run": os.path.basename(os.path.dirname(os.path.normpath(d))) or d,
           "commit": open(cfile).read().strip() if os.path.exists(cfile) else "n/a"}
    # cbca: one ITERATION over one volume = the launches of cbca_by_arms (tile instances + strip kernel, all but one of which
    # stand down at their first instruction): summed, not averaged
    groups = {"sgm": ("sgm_pass_kernel",), "cbca": ("cbca_strip_kernel", "cbca_tile_kernel", "cbca_lean_kernel", "cbca_lean2x_kernel", "cbca_classify_kernel", "cbca_classify2x_kernel", "cbca_list_kernel", "cbca_list_cost_kernel", "cbca_list_reset_kernel"),
              "join": ("join_owner_kernel",),
              "transpose": ("transpose_kernel", "transpose4_kernel")}
    for g, pat in groups.items():
        ks = [k for k in acc if any(q in k for q in pat)]
        if not ks:
            continue
        tot, per_kernel = 0.0, {}
        for k in ks:
            rd = 2 * sum(acc[k]["FETCH_SIZE"]) / max(1, len(acc[k]["FETCH_SIZE"])) * 1024
            wr = sum(acc[k]["WRITE_SIZE"]) / max(1, len(acc[k]["WRITE_SIZE"])) * 1024
            per_kernel[k[:90]] = dict(read=round(rd), write=round(wr))
            tot += rd + wr
        if g == "cbca":
            # bytes of ALL dispatches of the group / iterations; an iteration launches one kernel of each family (strip, tile<4, tile<13:
            # the plan-writing and plan-reading instances of a tile family are different kernels of the same family)
            # (textured pairs: cbca_lean2x_kernel per PAIR of iterations -- the strip kernel's stand-down launches still count the iterations --
            # and cbca_classify2x_kernel once per direction)
            fam = lambda k: ("strip" if "cbca_strip" in k else "lean" if "cbca_lean" in k else "once" if ("cbca_classify" in k or "cbca_list" in k) else
                             "tile4" if "cbca_tile_kernel<4," in k else "tile13")
            allb = sum((2 * sum(acc[k]["FETCH_SIZE"]) + sum(acc[k]["WRITE_SIZE"])) * 1024 for k in ks)
            iters = max(sum(len(acc[k]["FETCH_SIZE"]) for k in ks if fam(k) == f) for f in ("strip", "tile4", "tile13", "lean"))
            out[g] = round(allb / max(1, iters))
            out["cbca_iterations_seen"] = iters
        else:
            out[g] = round(tot / len(ks))
        out[g + "_kernels"] = per_kernel
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic_%s.json" % config)
    json.dump(out, open(path, "w"), indent=1)
    print(path, {k: v for k, v in out.items() if not k.endswith("_kernels") and k not in ("_note", "run", "commit")})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
