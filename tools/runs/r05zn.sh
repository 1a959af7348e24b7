echo "--- lazy"; ATLAS_TRACE=1 python tools/time_ra_lazy.py 2>&1 | grep -E "lazy rounds|ra_virtual d16" | sort | uniq -c | cut -c1-200
echo "--- gathered rows"; ATLAS_RA_LAZY_LOG=31 python tools/time_ra_lazy.py 2>&1 | grep "ra_virtual d16"
