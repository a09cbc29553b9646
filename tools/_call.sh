mkdir -p gpurun_out/r03t /tmp/c2bcli
for b in 32 128; do echo "\$ city2ba synthetic /tmp/c2bcli/g$b.bbal --blocks $b"; ( time C2B_TIMING=1 city2ba_amd/cli/city2ba synthetic /tmp/c2bcli/g$b.bbal --blocks $b ) 2>&1 | grep -v "^$\|user\|sys"; done > gpurun_out/r03t/cli_times.txt 2>&1
echo "\$ city2ba noise g128.bbal n128.bbal --drift-strength 1e-5 --rotation-std 0.01 --observation-std 0.001 --seed 1" >> gpurun_out/r03t/cli_times.txt
( time C2B_TIMING=1 city2ba_amd/cli/city2ba noise /tmp/c2bcli/g128.bbal /tmp/c2bcli/n128.bbal --drift-strength 1e-5 --rotation-std 0.01 --observation-std 0.001 --seed 1 ) 2>&1 | grep -v "^$\|user\|sys" >> gpurun_out/r03t/cli_times.txt
cat gpurun_out/r03t/cli_times.txt
