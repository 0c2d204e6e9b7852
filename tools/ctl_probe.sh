for V in 64 262144; do SURVEY_SEEDS=2,13,26,47,41,15,30 python tools/patch_survey.py 0 0 $V 24000 2>&1 | grep "^seed" | cut -c1-150; done
