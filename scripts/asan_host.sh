# Host-side AddressSanitizer / UBSan run of the C program (time loop, av ring, slab driver over sockets):
# the host C files are built with gcc -fsanitize=address,undefined, the kernels' object is the ordinary one
# (no device-side sanitizer), and a few program runs are repeated with that build; their outputs must equal
# the ordinary build's, and the sanitizers must stay silent.  usage (on a GPU box, from the repository root):
#   bash scripts/asan_host.sh [output directory]
set -e
R=$PWD; O=${1:-gpurun_out/asan}; mkdir -p $O; B=$O/build; mkdir -p $B
PKG=super-lattice-boltzmann-2d_amd
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
for f in slb_globals slb_bessel slb_cli slb_solver slb_main slb_link slb_slabs; do
  gcc -std=gnu99 $SAN -fPIC -ffp-contract=off -Wall -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $PKG/host/$f.c -o $B/$f.o
done
K=$PKG/csrc/slb_kernels.o          # (object files do not travel to a GPU box: built here then, without any sanitizer)
[ -f $K ] || { K=$B/slb_kernels.o; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -c $PKG/csrc/slb_kernels.hip -o $K 2> $B/hipcc.log; }
gcc $SAN -shared -fPIC -o $B/libslb2d.so $K $(ls $B/*.o | grep -v slb_kernels.o) -L/opt/rocm/lib -lamdhip64 -lstdc++ -lm -ldl -lpthread -Wl,-rpath,/opt/rocm/lib
gcc -std=gnu99 $SAN -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o $B/boltzmann_solver $PKG/host/boltzmann_solver.c -L$B -lslb2d -Wl,-rpath,$R/$B -Wl,-rpath,/opt/rocm/lib -lm
echo "sanitizer runtime linked: $(ldd $B/boltzmann_solver | grep -c -E "libasan|libubsan") libraries, $(nm -D $B/libslb2d.so | grep -c __asan_report) report entry points referenced by the library"
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0:log_path=$R/$O/asan
export UBSAN_OPTIONS=print_stacktrace=1:log_path=$R/$O/ubsan
C="E_dc=1 E_omega=0.1 mu=5 alpha=1 PhiYmin=-4 PhiYmax=4 dt=0.001 quiet=1"
run() {   # tag, then program arguments
  tag=$1; shift
  mkdir -p $O/$tag.ref $O/$tag.san
  (cd $O/$tag.ref && timeout -k 10 120 $R/bin/boltzmann_solver "$@" o=out.txt > stdout.txt 2> stderr.txt < ${STDIN:-/dev/null})
  (cd $O/$tag.san && timeout -k 10 300 $R/$B/boltzmann_solver "$@" o=out.txt > stdout.txt 2> stderr.txt < ${STDIN:-/dev/null}) || { echo "$tag: sanitizer build exited non-zero"; tail -5 $O/$tag.san/stderr.txt; }
  if diff -rq -x stderr.txt $O/$tag.ref $O/$tag.san > /dev/null; then echo "$tag: same output"; else echo "$tag: OUTPUT DIFFERS"; fi
}
run d4      display=4 omega=100 B=0.5 n-harmonics=16 g-grid=4001 t-max=0.03 $C
run d4wide  display=4 omega=100 B=0.0076 n-harmonics=32 g-grid=524288 t-max=0.012 $C
run d77     display=77 omega=100 B=0.12 n-harmonics=64 g-grid=32768 t-max=0.05 $C
run d8      display=8 omega=100 B=0.5 n-harmonics=8 g-grid=61 t-max=0.01 $C
run d7      display=7 omega=100 B=0.5 n-harmonics=6 g-grid=37 t-max=0.02 frame-start=0.03 $C
run d9      display=9 omega=100 B=0.5 n-harmonics=4 g-grid=13 t-max=0.02 $C
run d3      display=3 omega=100 B=0.5 n-harmonics=8 g-grid=61 t-max=0.01 $C
printf 'E_dc 1.5 0.02\nB 0.4 0.013\nomega 50 0.01\nbogus 1 0.005\nexit\n' > $O/records.txt
STDIN=$R/$O/records.txt run stdin display=4 omega=100 B=0.5 n-harmonics=8 g-grid=253 t-max=0.02 read-from=stdin $C
run r3d4    display=4 omega=100 B=0.5 n-harmonics=16 g-grid=4001 t-max=0.03 ranks=3 $C
SLB_STEPS_PER_LAUNCH=3 run r4d4t display=4 omega=40 B=0.8 n-harmonics=24 g-grid=5003 t-max=0.02 ranks=4 $C
run r3d77   display=77 omega=100 B=0.05 n-harmonics=16 g-grid=4001 t-max=0.05 ranks=3 $C
run r3d8    display=8 omega=100 B=0.5 n-harmonics=8 g-grid=61 t-max=0.01 ranks=3 $C
STDIN=$R/$O/records.txt run r3stdin display=4 omega=100 B=0.5 n-harmonics=8 g-grid=1001 t-max=0.02 read-from=stdin ranks=3 $C
echo "sanitizer reports:"; ls $O | grep -E "^(asan|ubsan)\." || echo "  none"
for f in $O/asan.* $O/ubsan.*; do [ -f "$f" ] && { echo "== $f"; head -40 "$f"; }; done 2>/dev/null | head -150
rm -rf $B $O/*.ref $O/*.san
