for r in 4 8 32 64; do
  for sc in "CornellBoxDiffuse 512 512 256 8" "LivingRoomLit 960 540 32 0" "MetalRings 960 540 32 0" "atrium 960 540 32 0" "clutter 960 540 32 0"; do
    set -- $sc
    MI_PT_LIB=$GRAFT_REPO_ROOT/master_amd/libmi_pt_r$r.so timeout -k 10 300 python bench.py --scene $1 --width $2 --height $3 --spp $4 --max-path $5 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/ab.json 2>/tmp/ab.err && python -c "
import json; d=json.load(open('/tmp/ab.json')); t=d['roofline']['terms']; print('R=$r %-20s %8.0f Ms/s N %6.2f T %5.2f N\' %6.2f T\' %5.2f'%('$1', d['value'], t['N'], t['T'], t['N_shadow_per_segment'], t['T_shadow_per_segment']))"
  done
done
