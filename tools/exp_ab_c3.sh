cp multi_part_assembly_amd/libmpa_hip.so /tmp/o.so
for rep in 1 2; do for v in a_old b_pk; do cp build_variants/$v.so multi_part_assembly_amd/libmpa_hip.so; python bench.py --config c3 --no-cpu-baseline --no-chamfer-standalone --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3))"; done; done
cp /tmp/o.so multi_part_assembly_amd/libmpa_hip.so
