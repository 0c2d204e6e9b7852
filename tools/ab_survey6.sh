# the survey's seeds named in SURVEY_SEEDS through several prebuilt libraries on ONE box, two alternating rounds (boxes differ by up to 25 % on single seeds)
cp s-rack_amd/libsrack_hip.so /tmp/keep.so
for round in 1 2; do for lib in "$@"; do
  cp $lib s-rack_amd/libsrack_hip.so
  echo "== $lib"
  python tools/patch_survey.py 0 1 262144 6000 2>/dev/null | grep "^seed" | sed -E 's/^seed +([0-9]+) .* ([0-9.]+) ms\/s.*/\1:\2/' | sort -t: -k1 -n | tr '\n' ' '; echo
done; done
cp /tmp/keep.so s-rack_amd/libsrack_hip.so
