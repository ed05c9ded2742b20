// dev (round 5): which kinds of load may a spin loop poll with?  One launch, two co-resident workgroups: workgroup 0 waits
// 20 us, then publishes a tagged 8-byte granule (agent-scope atomic store, like the pose mailbox / the tagged partial rows);
// workgroup 1 has READ THE LINE ONCE before (so its CU's vector L1 holds the old value) and then polls for the tag with
//   mode 0  a raw buffer load, aux = 0 (cached)
//   mode 1  a raw buffer load, aux = sc1 | volatile (0x80000010) — what search.hip's first look uses
//   mode 2  a volatile global load
//   mode 3  __hip_atomic_load(relaxed, agent scope) — what every re-poll uses since round 5
//   mode 4  mode 1 with a 16-byte load of two granules (solve_device.h::tagged_pair_load)
// each bounded by 2 ms of the 100 MHz wall clock.  Prints, per mode, whether the consumer saw the tag and after how long.
//   hipcc --offload-arch=gfx950 -O3 -o stale_poll.bin stale_poll.hip && ./stale_poll.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ inline long long wall() { return (long long)__builtin_amdgcn_s_memrealtime(); }

__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

template <int MODE>
__device__ inline unsigned long long look(const unsigned long long* p) {
    if constexpr (MODE == 0) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(p, 64), 0, 0, 0);
        return ((unsigned long long)(unsigned)v[1] << 32) | (unsigned)v[0];
    } else if constexpr (MODE == 1) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(p, 64), 0, 0, (int)0x80000010);
        return ((unsigned long long)(unsigned)v[1] << 32) | (unsigned)v[0];
    } else if constexpr (MODE == 4) {  // 16 bytes = two granules at once (tagged_pair_load); the second granule's tag decides
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(p, 64), 0, 0, (int)0x80000010);
        return (unsigned)v[1] == (unsigned)v[3] ? (((unsigned long long)(unsigned)v[3] << 32) | (unsigned)v[2]) : 0ull;
    } else if constexpr (MODE == 2) {
        return *(const volatile unsigned long long*)p;
    } else {
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int MODE>
__global__ __launch_bounds__(64) void k_poll(unsigned long long* box, unsigned tag, long long* out) {
    if (blockIdx.x == 0) {  // the producer
        if (threadIdx.x == 0) {
            const long long t0 = wall();
            while (wall() - t0 < 2000) __builtin_amdgcn_s_sleep(8);  // 20 us
            __hip_atomic_store(box, ((unsigned long long)tag << 32) | 42u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(box + 1, ((unsigned long long)tag << 32) | 43u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[2] = wall();
        }
        return;
    }
    if (threadIdx.x != 0) return;
    unsigned long long v = look<MODE>(box);  // the first look: the line is in this CU's L1 from here on
    const long long t0 = wall();
    int trips = 0;
    while ((unsigned)(v >> 32) != tag && wall() - t0 < 200000) {
        __builtin_amdgcn_s_sleep(2);
        v = look<MODE>(box);
        ++trips;
    }
    out[0] = (unsigned)(v >> 32) == tag ? wall() - t0 : -1;
    out[1] = trips;
}

template <int MODE>
static void run(const char* name) {
    unsigned long long* box;
    long long *out, h[3];
    hipMalloc(&box, 64);
    hipMalloc(&out, 24);
    for (unsigned tag = 1; tag <= 3; ++tag) {
        hipMemset(out, 0, 24);
        hipLaunchKernelGGL(k_poll<MODE>, dim3(2), dim3(64), 0, 0, box, tag, out);
        hipDeviceSynchronize();
        hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
        if (h[0] < 0) printf("%-44s tag %u: NEVER seen within 2 ms (%lld polls)\n", name, tag, h[1]);
        else printf("%-44s tag %u: seen after %.2f us (%lld polls)\n", name, tag, h[0] * 0.01, h[1]);
    }
    hipFree(box);
    hipFree(out);
}

int main() {
    run<0>("buffer_load aux=0 (cached)");
    run<1>("buffer_load aux=sc1|volatile");
    run<2>("volatile global load");
    run<3>("atomic load, relaxed, agent scope");
    run<4>("buffer_load b128 aux=sc1|volatile (pair)");
    return 0;
}
