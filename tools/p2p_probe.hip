// Feasibility probe for a one-shot exchange between two PROCESSES (one GPU here, peers over xGMI on a node): fine-grained device
// memory shared through hipIpc handles, every rank writes its 16 KB partial into every rank's inbox as 8-byte {value, sequence}
// words (a word arrives whole, so no fence and no separate flag), then sums the inbox slots in rank order once their sequence
// numbers match.  fork() happens before any HIP call.  hipcc --offload-arch=gfx950 -O3 -o tools/_build/p2p_probe tools/p2p_probe.hip
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <sys/wait.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s -> %s\n", rank, #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int N = 4128, WORLD = 2;  // 32 x 128 + 32 floats

struct Peers { unsigned long long* inbox[WORLD]; };

// inbox layout per rank: [parity 2][sender WORLD][N] words
__global__ void xchg(float* buf, Peers peers, int rank, unsigned seq, int* status) {
    const int parity = seq & 1;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int i = tid; i < N; i += nth) {
        const unsigned long long word = ((unsigned long long)seq << 32) | __float_as_uint(buf[i]);
        for (int p = 0; p < WORLD; ++p)
            __hip_atomic_store(peers.inbox[p] + ((size_t)parity * WORLD + rank) * N + i, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int i = tid; i < N; i += nth) {
        float sum = 0.f;
        for (int r = 0; r < WORLD; ++r) {
            const unsigned long long* src = peers.inbox[rank] + ((size_t)parity * WORLD + r) * N + i;
            unsigned long long w = 0;
            int spin = 0;
            for (; spin < (1 << 24); ++spin) {
                w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((unsigned)(w >> 32) == seq) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (spin == (1 << 24)) { *status = 1; return; }  // gave up: loud, not hung
            sum += __uint_as_float((unsigned)w);
        }
        buf[i] = sum;
    }
}

int main(int argc, char** argv) {
    const int fine = argc > 1 ? atoi(argv[1]) : 1, iters = argc > 2 ? atoi(argv[2]) : 200;
    int p01[2], p10[2];
    if (pipe(p01) || pipe(p10)) return 1;
    const pid_t pid = fork();
    const int rank = pid == 0 ? 1 : 0;
    CK(hipSetDevice(0));
    const size_t bytes = (size_t)2 * WORLD * N * sizeof(unsigned long long);
    unsigned long long* inbox = nullptr;
    if (fine) CK(hipExtMallocWithFlags((void**)&inbox, bytes, hipDeviceMallocFinegrained));
    else CK(hipMalloc((void**)&inbox, bytes));
    CK(hipMemset(inbox, 0, bytes));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t mine, theirs;
    CK(hipIpcGetMemHandle(&mine, inbox));
    if (rank == 0) { if (write(p01[1], &mine, sizeof mine) < 0 || read(p10[0], &theirs, sizeof theirs) < 0) return 1; }
    else { if (write(p10[1], &mine, sizeof mine) < 0 || read(p01[0], &theirs, sizeof theirs) < 0) return 1; }
    void* peer = nullptr;
    CK(hipIpcOpenMemHandle(&peer, theirs, hipIpcMemLazyEnablePeerAccess));
    Peers peers;
    peers.inbox[rank] = inbox;
    peers.inbox[1 - rank] = (unsigned long long*)peer;
    float *buf, *hbuf = (float*)malloc(N * sizeof(float));
    int* status;
    CK(hipMalloc(&buf, N * sizeof(float)));
    CK(hipMalloc(&status, sizeof(int)));
    CK(hipMemset(status, 0, sizeof(int)));
    int bad = 0;
    double us = 0;
    for (int k = 1; k <= iters; ++k) {
        for (int i = 0; i < N; ++i) hbuf[i] = (rank + 1) * 1.0f + i * 0.25f + k;
        CK(hipMemcpy(buf, hbuf, N * sizeof(float), hipMemcpyHostToDevice));
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(xchg, dim3(4), dim3(256), 0, 0, buf, peers, rank, (unsigned)k, status);
        CK(hipDeviceSynchronize());
        us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        CK(hipMemcpy(hbuf, buf, N * sizeof(float), hipMemcpyDeviceToHost));
        int st = 0;
        CK(hipMemcpy(&st, status, sizeof(int), hipMemcpyDeviceToHost));
        if (st) { fprintf(stderr, "rank %d: exchange %d gave up\n", rank, k); bad = 1; break; }
        for (int i = 0; i < N; ++i) {
            const float want = (1.0f + i * 0.25f + k) + (2.0f + i * 0.25f + k);
            if (hbuf[i] != want) { if (!bad) fprintf(stderr, "rank %d: exchange %d element %d: %g != %g\n", rank, k, i, hbuf[i], want); bad = 1; }
        }
        if (bad) break;
    }
    printf("rank %d: %s, %d exchanges, %.1f us per launch + sync (memory %s)\n", rank, bad ? "FAILED" : "ok", iters, us / iters,
           fine ? "fine-grained" : "coarse-grained");
    fflush(stdout);
    CK(hipIpcCloseMemHandle(peer));
    if (rank == 0) { int wst = 0; waitpid(pid, &wst, 0); return bad || wst; }
    return bad;
}
