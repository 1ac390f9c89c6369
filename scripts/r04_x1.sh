O=gpurun_out/r04_x1; mkdir -p $O
L=$PWD/cuopt_amd/lib
python scripts/r04_x1.py '[
 ["row", "c3", {"CUOPT_AMD_TUNE": "panel_seg=0"}],
 ["row_nosum", "c3", {"CUOPT_AMD_TUNE": "panel_seg=0", "CUOPT_AMD_LIB": "'$L'/libx_X_ROW_NOSUM.so"}],
 ["seg", "c3", {"CUOPT_AMD_TUNE": "panel_seg=1"}],
 ["seg_noemit", "c3", {"CUOPT_AMD_TUNE": "panel_seg=1", "CUOPT_AMD_LIB": "'$L'/libx_X_SEG_NOEMIT.so"}],
 ["seg_nosum", "c3", {"CUOPT_AMD_TUNE": "panel_seg=1", "CUOPT_AMD_LIB": "'$L'/libx_X_SEG_NOSUM.so"}],
 ["row", "powerlaw", {"CUOPT_AMD_TUNE": "panel_seg=0"}],
 ["seg", "powerlaw", {"CUOPT_AMD_TUNE": "panel_seg=1"}],
 ["seg_nosum", "powerlaw", {"CUOPT_AMD_TUNE": "panel_seg=1", "CUOPT_AMD_LIB": "'$L'/libx_X_SEG_NOSUM.so"}]
]' 2>&1 | tee $O/table.txt
