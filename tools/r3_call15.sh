set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/phase_profile.py cfg5_sphere_cloud_4k 16 gpurun_out/executed_cfg5_r3o.json > gpurun_out/phases_cfg5_r3o.txt
cat gpurun_out/phases_cfg5_r3o.txt
RT_DEEP_BINARY=1 timeout -k 10 200 python tools/phase_profile.py cfg5_sphere_cloud_4k 16 gpurun_out/executed_cfg5_r3o_binary.json > gpurun_out/phases_cfg5_r3o_binary.txt
cat gpurun_out/phases_cfg5_r3o_binary.txt
