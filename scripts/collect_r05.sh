set -o pipefail
cd $GRAFT_REPO_ROOT
bash scripts/collect_profiles.sh r05_b > gpurun_out/r05_b_collect.log 2>&1 &&
python3 scripts/heads_time.py --stamps > gpurun_out/r05_b/heads_fused_stamps.txt 2>&1 &&
python3 scripts/heads_time.py > gpurun_out/r05_b/heads_conv5_time.txt 2>&1 &&
python3 scripts/dec_time.py > gpurun_out/r05_b/dec_fused_time.txt 2>&1 &&
python3 scripts/knn_time.py > gpurun_out/r05_b/knn_feat_forms_time.txt 2>&1 &&
python3 scripts/knn_serial_rate.py > gpurun_out/r05_b/knn_serial_rate.txt 2>&1 &&
echo all-done
