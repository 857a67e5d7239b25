for rep in 1 2; do
for cfg in "libmi_pt.so 12" "libmi_pt.so 8" "libmi_pt_dyn7.so 8"; do set -- $cfg; lib=$1; rows=$2
for spec in "atrium 1920 1080 128" "clutter 1920 1080 64" "atrium:2000000 1920 1080 32" "LivingRoomLit 1920 1080 64"; do set -- $spec
MI_PT_STACK_ROWS=$rows MI_PT_LIB=$PWD/master_amd/$lib python bench.py --scene $1 --width $2 --height $3 --spp $4 --max-path 0 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse --no-fast-variant --no-live-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rows=$rows %-18s %8.1f Msamples/s  lds %d' % ('$1', d['value'], d['config']['launch']['lds_bytes_per_workgroup']))"; done; done; done
