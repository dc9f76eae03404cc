# Copies the files one run of tools/profile_round.sh <tag> left under gpurun_out/ into profiles/ under the names the
# design notes cite.  usage: bash tools/install_evidence.sh <tag> <name-prefix>   (e.g. r3z r3)
set -e
T=$1; N=$2; G=gpurun_out; P=profiles
cp $G/prof_$T.json $P/${N}_bench_cfg2.json
cp $G/prof_nobase_$T.json $P/${N}_bench_cfg2_profiled_command.json     # the line of the run the kernel stats were taken from
cp $G/${T}_bench_all_workloads_64spp.jsonl $P/${N}_bench_all_workloads_64spp.jsonl
for f in $G/${T}_fused_*kernel_stats.csv; do b=$(basename $f); cp $f $P/${N}_${b#${T}_}; done
cp $G/bench_wavefront_cfg2_$T.json $P/${N}_bench_wavefront_cfg2_64spp.json
cp $G/${T}_wavefront_cfg2_64spp_kernel_stats.csv $P/${N}_wavefront_cfg2_64spp_kernel_stats.csv
cp $G/wavefront_traffic_cfg2_random_spheres_1080p_$T.json $P/${N}_wavefront_traffic_cfg2_64spp.json
cp $G/bench_small_launches_$T.jsonl $P/${N}_bench_small_launches.jsonl
cp $G/bench_cfg5_full_$T.json $P/${N}_bench_cfg5_full_1024spp_per_gpu.json
cp $G/bench_cfg5_64spp_$T.json $P/${N}_bench_cfg5_64spp.json
cp $G/bench_cfg5_binary_$T.json $P/${N}_bench_cfg5_64spp_two_children_nodes.json
cp $G/bench_cfg5_plain_$T.json $P/${N}_bench_cfg5_64spp_two_children_nodes_without_lds_top.json
cp $G/interactive_fps_$T.txt $P/${N}_interactive_fps.txt
for w in cfg2_random_spheres_1080p cfg3_cornell_smoke_1080p cfg4_csg_textured_1080p cfg5_sphere_cloud_4k; do
  cp $G/pmc_sq_${w}_$T.json $P/${N}_pmc_sq_$w.json
  cp $G/pmc_sq_${w}_$T.json $P/pmc_sq_$w.json          # the copy bench.py quotes
  cp $G/pmc_mem_${w}_$T.json $P/${N}_pmc_mem_$w.json
  cp $G/phases_${w}_$T.txt $P/${N}_phases_${w}_16spp.txt
  cp $G/executed_${w}_$T.json $P/executed_$w.json
done
cp $G/traffic_record_cfg2_random_spheres_1080p_$T.json $P/${N}_traffic_cfg2_random_spheres_1080p_1024spp.json
cp $G/traffic_record_cfg2_random_spheres_1080p_$T.json $P/traffic_cfg2_random_spheres_1080p.json
cp $G/traffic_record_cfg5_sphere_cloud_4k_$T.json $P/${N}_traffic_cfg5_sphere_cloud_4k_16spp.json
python tools/resource_usage.py > $P/${N}_kernel_resource_usage.txt 2>/dev/null
bash tools/count_packed.sh > $P/${N}_packed_fp32_instructions.txt 2>/dev/null
