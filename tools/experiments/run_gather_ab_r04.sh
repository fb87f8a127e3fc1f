for v in base g8 base g8; do
  if [ $v = base ]; then unset CDML_LIB_PATH; else export CDML_LIB_PATH=$PWD/build/variants/libcdml_$v.so; fi
  python bench.py --steps 60 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);print('$v',d['ms_per_step'],d['gather']['achieved'],d['gather']['frac'],d['gather']['launch_ms'])"
done
