p='lld_slam_amd/csrc/lld_orb_search.hip'
s=open(p).read()
s=s.replace("  // ---------------------------------------------------------------- keypoints into LDS (+ grid counting sort)\n","  const long long tc0 = wall_clock64();\n  // ---------------------------------------------------------------- keypoints into LDS (+ grid counting sort)\n",1)
s=s.replace("  // ---------------------------------------------------------------- fixed-point rounds, one lane per query\n","  const long long tc1 = wall_clock64(); long long tr1 = 0;\n  // ---------------------------------------------------------------- fixed-point rounds, one lane per query\n",1)
s=s.replace("    const int changed = ctl[0];\n","    const int changed = ctl[0]; if (rounds == 1) tr1 = wall_clock64();\n",1)
s=s.replace("  // ---------------------------------------------------------------- rotation histogram, owners, counts\n","  const long long tc2 = wall_clock64();\n  // ---------------------------------------------------------------- rotation histogram, owners, counts\n",1)
s=s.replace("  if (tid == 0) { P.summary[0] = ctl[1] - ctl[2]; P.summary[1] = rounds; }","  if (tid == 0) { P.summary[0] = ctl[1] - ctl[2]; P.summary[1] = rounds; if (blockIdx.x == 0) printf(\"PH setup %lld round1 %lld later %lld (n=%d) final %lld [10ns]\\n\", tc1 - tc0, tr1 - tc1, tc2 - tr1, rounds, wall_clock64() - tc2); }")
open(p,'w').write(s)
