set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
RT_SHOW_OCCUPANCY=1 timeout -k 10 200 python bench.py --workload cfg5_sphere_cloud_4k --spp 8 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r3m_show.json 2> gpurun_out/r3m_show.err
sort gpurun_out/r3m_show.err | uniq -c | head
