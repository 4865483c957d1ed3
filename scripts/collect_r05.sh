set -o pipefail
cd $GRAFT_REPO_ROOT
TAG=${1:-r05_c}
bash scripts/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1 &&
python3 scripts/heads_time.py --stamps > gpurun_out/$TAG/heads_fused_stamps.txt 2>&1 &&
python3 scripts/heads_time.py > gpurun_out/$TAG/heads_conv5_time.txt 2>&1 &&
python3 scripts/dec_time.py > gpurun_out/$TAG/dec_fused_time.txt 2>&1 &&
python3 scripts/knn_time.py > gpurun_out/$TAG/knn_feat_forms_time.txt 2>&1 &&
python3 scripts/knn_serial_rate.py > gpurun_out/$TAG/knn_serial_rate.txt 2>&1 &&
python3 scripts/heads_time.py --knobs 0,1,2,4,8,16 > gpurun_out/$TAG/heads_conv5_knobs.txt 2>&1 &&
python3 scripts/hs_chain_time.py --knobs 0,1,3,4 > gpurun_out/$TAG/hs_chain_time.txt 2>&1 &&
python3 scripts/dec_l1_time.py --knobs 0,1,2,3 > gpurun_out/$TAG/dec_l1_time.txt 2>&1 &&
(cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_valu $GRAFT_REPO_ROOT/scripts/micro/mfma_valu_overlap.hip && /tmp/mfma_valu > $GRAFT_REPO_ROOT/gpurun_out/$TAG/mfma_valu_overlap.txt 2>&1) &&
python3 scripts/proj_time.py --knobs 0,1 > gpurun_out/$TAG/proj_time.txt 2>&1 &&
bash scripts/env_ab.sh 2 TGP_HS_CHAIN=0 TGP_HS_CHAIN=1 > gpurun_out/$TAG/hs_chain_ab.txt 2>&1 &&
bash scripts/env_ab.sh 3 TGP_PROJ_KERNEL=0 TGP_PROJ_KERNEL=1 > gpurun_out/$TAG/proj_kernel_ab.txt 2>&1 &&
echo all-done
