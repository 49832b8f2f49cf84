cd $GRAFT_REPO_ROOT
for a in "" "dle_cfg=1" "dle_s2=2" "dle_s2=4" "dle_s2=6" "dle_s1=2" "dle_s1=3 dle_s2=4" "dle_units=16" "dle_units=48" "dle_frac_a=60" "dle_frac_a=85" "" ; do
  echo -n "[$a] "; timeout 120 python tools/dle_stats.py $a 2>&1 | grep -v amdgpu.ids | cut -c1-60
done
