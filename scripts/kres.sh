#!/bin/bash
# print per-kernel register/scratch usage of a .hip file
cd /root/repo/i2sdf_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include $KRES_EXTRA -x hip -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - 
