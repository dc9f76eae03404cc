# HBM traffic of the wavefront pipeline (all of its kernels: ray generation, extend, shade, tail, accumulate) per render
# pass, from the L2 fabric counters, one rocprofv3 pass per counter.
# usage (GPU box): bash tools/wavefront_traffic.sh <workload> <spp> <tag>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-cfg2_random_spheres_1080p}; SPP=${2:-64}; TAG=${3:-cur}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/wfpmc_${c}_$TAG
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/wfpmc_${c}_$TAG -- python3 bench.py --workload $W --pipeline wavefront --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/wfpmc_${c}_$TAG.json 2> gpurun_out/wfpmc_${c}_$TAG.err
done
python3 - $W $SPP $TAG <<'PY' > gpurun_out/wavefront_traffic_${1:-cfg2_random_spheres_1080p}_${3:-cur}.json
import csv, glob, json, sys
w, spp, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
out = {"workload": w, "spp": spp, "passes": 3}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = {}
    for path in glob.glob("gpurun_out/wfpmc_%s_%s/**/*counter_collection.csv" % (c, tag), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]
            if name.startswith("k_wf_"):
                per[name] = per.get(name, 0.0) + float(row["Counter_Value"])
    # the counters are in kilobytes (FETCH_SIZE / WRITE_SIZE of the L2's fabric side)
    out[c + "_KB_per_pass_by_kernel"] = {k: v / 3 for k, v in sorted(per.items())}
    out[c + "_GB_per_pass"] = sum(per.values()) / 3 * 1024 / 1e9
# on gfx950 FETCH_SIZE counts half of the fetched bytes: doubled; WRITE_SIZE taken as is
out["hbm_GB_per_pass"] = 2.0 * out["FETCH_SIZE_GB_per_pass"] + out["WRITE_SIZE_GB_per_pass"]
line = json.load(open("gpurun_out/wfpmc_FETCH_SIZE_%s.json" % tag))
out["ms_per_pass_under_the_profiler"] = line["ms_per_step"]
print(json.dumps(out, indent=1))
PY
cat gpurun_out/wavefront_traffic_${W}_$TAG.json
