cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%9.1f fps %.4f ms plain %s' % (d['value'], d['ms_per_step'], d.get('plain_forward_detect_fps')))"; }
cp achelous_amd/libachelous_hip.so /tmp/keep.so
for rep in 1 2; do
  echo "default          $(run)"
  echo "spp_split=3      $(run --opt spp_split=3)"
  echo "spp_split=6      $(run --opt spp_split=6)"
  echo "band_rows_s3=3   $(run --opt band_rows_s3=3)"
  echo "band_rows_s3=2   $(run --opt band_rows_s3=2)"
  cp achelous_amd/csrc/build/variants/lib_sdta512.so achelous_amd/libachelous_hip.so; echo "sdta512          $(run)"
  cp achelous_amd/csrc/build/variants/lib_sdta256.so achelous_amd/libachelous_hip.so; echo "sdta256          $(run)"
  cp /tmp/keep.so achelous_amd/libachelous_hip.so
done
