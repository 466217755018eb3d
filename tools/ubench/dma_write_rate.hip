// What a file writer of the pipe sees: write() of a page-locked buffer that a D2H copy has JUST filled, two files at once (the two
// good files of a paired run), against write() of a buffer that has been lying still — on either NUMA node, in pieces of 1 / 4 / 16 / 45
// MiB, with and without fallocate.  (round-5 review: the pipe's writers ran at 8.7 GB/s per file where write() alone measures
// 11.8 - 12: is it the source buffer — DMA-fresh lines, the other socket — or the piece size?)
//   hipcc -O2 -o dma_write_rate dma_write_rate.hip -lpthread ;  dma_write_rate DIR [GiB per file]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool bind_node(int node) {
    char path[96], list[4096] = "";
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    const bool got = fgets(list, sizeof(list), f) != nullptr;
    fclose(f);
    if (!got) return false;
    cpu_set_t want; CPU_ZERO(&want);
    for (char* p = list; *p;) {
        char* q; long a = strtol(p, &q, 10); if (q == p) break; long b = a;
        if (*q == '-') { p = q + 1; b = strtol(p, &q, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET((int)c, &want);
        p = *q == ',' ? q + 1 : q; if (*q != ',') break;
    }
    return sched_setaffinity(0, sizeof(want), &want) == 0;
}
// page-locked the way the pipe does it (aqc_host_alloc): anonymous huge pages, touched on `node`, registered
static uint8_t* pinned_on(int node, size_t n) {
    uint8_t* p = nullptr;
    std::thread t([&] {
        if (node >= 0) bind_node(node);
        p = (uint8_t*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(p, n, MADV_HUGEPAGE);
        memset(p, 'A', n);
    });
    t.join();
    if (hipHostRegister(p, n, hipHostRegisterPortable) != hipSuccess) { fprintf(stderr, "hipHostRegister failed\n"); exit(1); }
    return p;
}

struct Case { const char* what; bool dma; int buf_node, writer_node; size_t piece; int falloc; };      // falloc: 0 no, 1 the whole file at once (posix_fallocate), 2 KEEP_SIZE 1 GiB ahead, 3 size-extending 1 GiB ahead

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : "/tmp";
    const size_t per_file = (size_t)((argc > 2 ? atof(argv[2]) : 1.7) * (1u << 30));
    const size_t CHUNK = 45u << 20;                      // what a chunk of 131072 pairs gives a good file
    const int n_nodes = access("/sys/devices/system/node/node1/cpulist", R_OK) == 0 ? 2 : 1;
    int gpu_node = 0;
    {
        char bus[64] = "";
        if (hipDeviceGetPCIBusId(bus, sizeof(bus), 0) == hipSuccess) {
            for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
            char path[160]; snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
            if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &gpu_node) != 1 || gpu_node < 0) gpu_node = 0; fclose(f); }
        }
    }
    printf("%d NUMA node(s), the GPU hangs off node %d; two files of %.2f GiB at once, one writer thread each, chunks of 45 MiB\n", n_nodes, gpu_node, per_file / 1073741824.0);
    uint8_t* dev = nullptr;
    if (hipMalloc(&dev, 2 * CHUNK) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    (void)hipMemset(dev, 'B', 2 * CHUNK);
    std::vector<Case> cases;
    const int other = n_nodes > 1 ? 1 - gpu_node : gpu_node;
    cases.push_back({"still buffer, write() of whole chunks", false, gpu_node, gpu_node, CHUNK, false});
    cases.push_back({"DMA-fresh buffer (D2H just landed), whole chunks", true, gpu_node, gpu_node, CHUNK, false});
    cases.push_back({"DMA-fresh, pieces of 16 MiB", true, gpu_node, gpu_node, 16u << 20, false});
    cases.push_back({"DMA-fresh, pieces of 4 MiB", true, gpu_node, gpu_node, 4u << 20, false});
    cases.push_back({"DMA-fresh, pieces of 1 MiB", true, gpu_node, gpu_node, 1u << 20, false});
    cases.push_back({"DMA-fresh, whole chunks, fallocate first", true, gpu_node, gpu_node, CHUNK, 1});
    cases.push_back({"DMA-fresh, whole chunks, KEEP_SIZE fallocate 1 GiB ahead", true, gpu_node, gpu_node, CHUNK, 2});
    cases.push_back({"DMA-fresh, whole chunks, size-extending fallocate 1 GiB ahead", true, gpu_node, gpu_node, CHUNK, 3});
    if (n_nodes > 1) {
        cases.push_back({"DMA-fresh, buffer on the GPU's node, writer on the other", true, gpu_node, other, CHUNK, false});
        cases.push_back({"DMA-fresh, buffer on the other node, writer on the GPU's", true, other, gpu_node, CHUNK, false});
        cases.push_back({"DMA-fresh, buffer and writer on the other node", true, other, other, CHUNK, false});
        cases.push_back({"still buffer, buffer and writer on the other node", false, other, other, CHUNK, false});
    }
    cases.push_back({"DMA-fresh, writer not bound", true, gpu_node, -1, CHUNK, false});
    for (int rep = 0; rep < 2; ++rep)
        for (const Case& c : cases) {
            double rate[2] = {0, 0};
            std::vector<std::thread> th;
            for (int f = 0; f < 2; ++f)
                th.emplace_back([&, f] {
                    (void)hipSetDevice(0);
                    uint8_t* buf[2] = {pinned_on(c.buf_node, CHUNK), pinned_on(c.buf_node, CHUNK)};      // two sets, like a slot worker's
                    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
                    if (c.writer_node >= 0) bind_node(c.writer_node);
                    char path[256]; snprintf(path, sizeof(path), "%s/dmaw_%d.bin", dir, f);
                    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
                    if (c.falloc == 1) (void)posix_fallocate(fd, 0, (off_t)per_file);
                    size_t reserved = 0;
                    double t_write = 0;
                    size_t done = 0;
                    int set = 0;
                    // the copy of chunk k + 1 runs while chunk k is written (the pipe's slots do the same)
                    if (c.dma) (void)hipMemcpyAsync(buf[0], dev + f * CHUNK, CHUNK, hipMemcpyDeviceToHost, st);
                    while (done < per_file) {
                        if (c.dma) {
                            (void)hipStreamSynchronize(st);
                            (void)hipMemcpyAsync(buf[1 - set], dev + f * CHUNK, CHUNK, hipMemcpyDeviceToHost, st);
                        }
                        const double t0 = now();
                        if (c.falloc >= 2 && done + CHUNK + (256u << 20) > reserved) {
                            (void)fallocate(fd, c.falloc == 2 ? FALLOC_FL_KEEP_SIZE : 0, (off_t)reserved, (off_t)(1u << 30));
                            reserved += 1u << 30;
                        }
                        for (size_t o = 0; o < CHUNK;) {
                            const ssize_t w = write(fd, buf[set] + o, std::min(c.piece, CHUNK - o));
                            if (w <= 0) { perror("write"); exit(1); }
                            o += (size_t)w;
                        }
                        t_write += now() - t0;
                        done += CHUNK;
                        set = 1 - set;
                    }
                    (void)hipStreamSynchronize(st);
                    if (c.falloc) (void)ftruncate(fd, (off_t)done);
                    close(fd);
                    unlink(path);
                    rate[f] = done / t_write / 1e9;
                    for (auto b : buf) { (void)hipHostUnregister(b); munmap(b, CHUNK); }
                    (void)hipStreamDestroy(st);
                });
            for (auto& t : th) t.join();
            printf("%-62s %6.2f + %6.2f GB/s inside write()\n", c.what, rate[0], rate[1]);
            fflush(stdout);
        }
    (void)hipFree(dev);
    return 0;
}
