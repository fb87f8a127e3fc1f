# Round 4: the last round of the plane-output products as 128 x 256 half tiles -- CDML_X3_HALFTILES=1 (the default: full and
# half tiles in ONE launch), =2 (the half tiles as a launch of their own), =0 (full tiles only); one box, alternating
# processes; fc1m / dh1m at config 1 (640 tiles = 2.5 rounds), and at config 3's per-GPU rows (24576: 1920 tiles = 7.5 rounds).
for v in 0 1 2 0 1 2; do
  echo "== CDML_X3_HALFTILES=$v"
  CDML_X3_HALFTILES=$v python tools/x3_gemm_probe.py --cases fc1m,dh1m --rounds 3 2>&1 | grep -v amdgpu.ids
  CDML_X3_HALFTILES=$v python tools/x3_gemm_probe.py --cases fc1m,dh1m --rounds 3 --rows 24576 2>&1 | grep -v amdgpu.ids | sed 's/^/rows 24576: /'
done
