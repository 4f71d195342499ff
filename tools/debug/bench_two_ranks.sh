cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3_b2
for mode in eager graph; do
timeout 240 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29547 bench.py --gpus 2 --steps 2 --warmup 1 --size 64 --batch 4 --backend gloo --all-on-device0 --mode $mode > gpurun_out/r3_b2/out_$mode.txt 2> gpurun_out/r3_b2/err_$mode.txt; echo "$mode rc $?"; tail -c 300 gpurun_out/r3_b2/out_$mode.txt; grep -v "^W\|^$" gpurun_out/r3_b2/err_$mode.txt | tail -25
done
