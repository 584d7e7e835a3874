// rxr_jitc -- the background compiler of program sets (RXR_SHADER_JIT=async, rxr_jit.hip) as a PROCESS: hiprtc cannot be joined
// from, or abandoned under, a library's static destructors, but a child process can simply be killed.  The compiler itself stays
// in the library: this program loads it and calls rxr_debug_jit_compile_file.
//   usage: rxr_jitc <librxr_hip.so> <generated source> <arch> <template level> <code object out>
#include <dlfcn.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <thread>

int main(int argc, char **argv) {
    if (argc != 6) {
        std::fprintf(stderr, "usage: %s <librxr_hip.so> <source> <arch> <level> <out>\n", argv[0]);
        return 2;
    }
    // a parent that dies without dropping its job takes this process with it (a watching thread rather than PR_SET_PDEATHSIG:
    // that signal follows the parent's spawning THREAD, and a render call may come from a pool thread that ends early)
    const pid_t parent = getppid();
    std::thread([parent] {
        while (getppid() == parent) usleep(200 * 1000);
        _Exit(6);
    }).detach();
    void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        std::fprintf(stderr, "rxr_jitc: %s\n", dlerror());
        return 3;
    }
    typedef int (*compile_fn)(const char *, const char *, int, const char *);
    compile_fn compile = (compile_fn)dlsym(lib, "rxr_debug_jit_compile_file");
    if (!compile) {
        std::fprintf(stderr, "rxr_jitc: the library has no rxr_debug_jit_compile_file\n");
        return 4;
    }
    const int rc = compile(argv[2], argv[3], std::atoi(argv[4]), argv[5]);
    // (the parent removes the job's directory -- source, code object, the compiler's temporaries -- when it collects or drops the job)
    std::fflush(nullptr);
    _Exit(rc == 0 ? 0 : 5);  // (no static destructors: the process has done its one job)
}
