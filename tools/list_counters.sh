#!/bin/bash
# counters rocprofv3 offers on this GPU whose names match the pattern (default: address translation, L2 / fabric stalls)
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "${1:-utcl|tlb|TCC_BUSY|TCC_TAG_STALL|TCC_EA0_RDREQ|TCC_EA0_WRREQ|TCC_REQ|TCC_HIT|TCC_MISS|MALL|DRAM}" | grep -i "name" | sed 's/^[ \t]*//' | sort -u | head -150
