// launcher.cpp — the `swarm` executable: sets what must be set before the OpenMP runtime is loaded, then loads
// libswarm_amd.so (next to it: ../lib) and runs the command line in it (swa_cli_main, host/main.cpp).  libgomp reads
// OMP_WAIT_POLICY in its constructor, i.e. before main of anything that links it — hence nothing here links it.
#include <dlfcn.h>
#include <libgen.h>
#include <unistd.h>

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <string>

int main(int argc, char ** argv) {
  setenv("OMP_WAIT_POLICY", "passive", 0);                  // (the user's own setting stands)
  std::string lib = "libswarm_amd.so";
  char self[PATH_MAX];
  const ssize_t len = readlink("/proc/self/exe", self, sizeof(self) - 1);
  if (len > 0) {
    self[len] = '\0';
    lib = std::string(dirname(self)) + "/../lib/libswarm_amd.so";
  }
  if (const char * env = std::getenv("SWARM_AMD_LIB")) { lib = env; }
  void * handle = dlopen(lib.c_str(), RTLD_NOW | RTLD_GLOBAL);
  if (handle == nullptr) { std::fprintf(stderr, "\nError: %s\n", dlerror()); return EXIT_FAILURE; }
  using cli_main = int (*)(int, char **);
  const auto run = reinterpret_cast<cli_main>(dlsym(handle, "swa_cli_main"));
  if (run == nullptr) { std::fprintf(stderr, "\nError: %s lacks swa_cli_main\n", lib.c_str()); return EXIT_FAILURE; }
  return run(argc, argv);
}
