bash tools/r05_search_pmc.sh r05pmc2
bash tools/r05_quick.sh r05g "tests/test_icp_gpu.py::test_wave_search_equals_the_per_query_walks" "wave_one:SMHIP_WAVE_SEARCH=1,no_overlap=1;lds_one:SMHIP_WAVE_SEARCH=0,no_overlap=1"
