set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/layer_roofline.py 30 > gpurun_out/layer_roofline.md 2> gpurun_out/layer_roofline.err
tail -3 gpurun_out/layer_roofline.err
timeout 500 python tools/soak.py DER resnet18 4 6 > gpurun_out/soak_der.txt 2>&1
tail -12 gpurun_out/soak_der.txt
