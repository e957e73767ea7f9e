#!/bin/bash
# per-kernel register / LDS / occupancy report for one csrc file:  tools/kres.sh mlp3
cd /root/repo/contextgs_amd/csrc
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -Rpass-analysis=kernel-resource-usage -c $1.hip -o /tmp/$1.o 2>&1 | grep -E "error|Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: //;s/\[-Rpass.*//' | paste - - - - - | sed -E 's/Function Name: _Z[0-9]+([a-z0-9_]+)I([A-Za-z0-9_]+).*VGPRs: ([0-9]+).*ScratchSize \[bytes\/lane\]: ([0-9]+).*Occupancy \[waves\/SIMD\]: ([0-9]+).*LDS Size \[bytes\/block\]: ([0-9]+).*/\1 \2 vgpr=\3 scratch=\4 occ=\5 lds=\6/'
