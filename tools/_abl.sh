for w in 0 1 2; do echo "WPAT $w"; MSGL_M256_WPAT=$w MSGL_M256_ABLATE=35 timeout 60 python tools/_steps.py 2>&1 | grep ABL; done
