/* boltzmann_solver -- program entry; everything lives in libslb2d.so */
#include "slb_host.h"

int main(int argc, char **argv) { return slb_main(argc, argv); }
