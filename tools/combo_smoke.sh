cd data
run() { echo "== $*"; timeout 120 "$@" 2>&1 | grep -E "^[0-9]+ [0-9]|ERROR|rror|Warning|PERF_SUMMARY$" | tail -2; echo "rc=$?"; }
run ../minimd_amd/bin/miniMD_dp -i in.eam.miniMD -s 8 --half_neigh 1 -n 40
run ../minimd_amd/bin/miniMD_dp -s 8 --half_neigh -1 -n 40
run ../minimd_amd/bin/miniMD_sp -s 8 --half_neigh 1 -gn 0 -n 40
run ../minimd_amd/bin/miniMD_dp -s 3 -n 40 --half_neigh 0
run ../minimd_amd/bin/miniMD_dp -s 3 -n 40 --half_neigh 1
run ../minimd_amd/bin/miniMD_dp -nx 2 -ny 2 -nz 20 -n 40 --half_neigh 0
run ../minimd_amd/bin/miniMD_dp -nx 2 -ny 2 -nz 20 -n 40 --half_neigh 1
run ../minimd_amd/bin/miniMD_dp -s 12 -b 1 -n 40 --half_neigh 0
run ../minimd_amd/bin/miniMD_dp -s 12 -b 30 -n 40 --half_neigh 1
run ../minimd_amd/bin/miniMD_dp -s 12 -b 30 -n 40 --half_neigh 0
run ../minimd_amd/bin/miniMD_dp -s 8 --ntypes 8 -n 40 --half_neigh 1
run ../minimd_amd/bin/miniMD_dp -s 8 --sort 0 -n 60 --half_neigh 1
run ../minimd_amd/bin/miniMD_dp -s 8 --sort 3 -n 60 --half_neigh 0 --check_exchange
run ../minimd_amd/bin/miniMD_sp -i in.eam.miniMD -s 6 -n 40 --half_neigh 0
