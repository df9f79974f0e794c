cd /root/repo
for t in "" 8,1 4,1 2,1 1,1 4,2 2,2 1,2; do
  for w in fwd dgrad; do
    echo -n "conv3 cfg[$t] "; CLHIP_CONV3_CFG=$t python tools/conv_micro.py 256 4 4 512 512 3 1 $w 50 2>/dev/null
  done
done
echo "--- layer3 8x8 256"
for t in "" 8,1 4,1 2,1 1,1 4,2 2,2 1,2; do
  for w in fwd dgrad; do
    echo -n "conv3 cfg[$t] "; CLHIP_CONV3_CFG=$t python tools/conv_micro.py 256 8 8 256 256 3 1 $w 50 2>/dev/null
  done
done
