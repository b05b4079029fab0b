// Heuristic-table kernel variants side by side on the 180 x 60 m map at position_resolution 10 (1800 x 600 x 73
// entries, turning radius 0.35): time per launch (HIP events, median of `reps`) and whether the outputs are equal
// bit for bit.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o profiles/table_micro profiles/table_micro.hip
//   walk      rs::solve_hoisted: the 44 candidates in reference order through one out-of-line formula switch, grid-stride loop
//   regroup   rs::shortest_length, same launch shape
//   regroup1  rs::shortest_length, one entry per thread
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../path_planner_amd/csrc/hyas_rs.h"

using namespace hyas;

#define CHECK(x)                                                    \
    do                                                              \
    {                                                               \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess)                                       \
        {                                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

template <int FORM>
__global__ void __launch_bounds__(256) k_table(const float *__restrict__ xs, uint32_t nx, const float *__restrict__ ys, uint32_t ny,
                                               const float *__restrict__ ts, uint32_t nt, double rho, float *__restrict__ out)
{
    __shared__ double s_qa[8];
    if (FORM != 0)
    {
        const double qa[8] = HY_QUARTER_ATANS;
        if (threadIdx.x < 8)
            s_qa[threadIdx.x] = qa[threadIdx.x];
        __syncthreads();
    }
    const uint32_t total = nt * ny * nx;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x)
    {
        const uint32_t i = e % nx, j = (e / nx) % ny, k = e / (nx * ny);
        const double q0[3] = {0, 0, 0};
        const double q1[3] = {(double)xs[i], (double)ys[j], (double)ts[k]};
        double x, y, phi;
        rs::normalise(q0, q1, rho, x, y, phi);
        if (FORM == 0)
            out[e] = (float)(rho * rs::solve_hoisted(x, y, phi).total);
        else
            out[e] = (float)(rho * rs::shortest_length(x, y, phi, s_qa));
    }
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    const float pr = argc > 2 ? (float)atof(argv[2]) : 10.f;
    const int forms = argc > 3 ? atoi(argv[3]) : 7; // bit mask of the forms to run
    std::vector<float> xs, ys, ts;
    for (float x = 0; x < 180.f; x += 1.0f / pr)
        xs.push_back(x);
    for (float y = 0; y < 60.f; y += 1.0f / pr)
        ys.push_back(y);
    const float dh = (float)(2 * HY_PI / 72.0);
    for (float th = 0; (double)th < 2 * HY_PI; th += dh)
        ts.push_back(th);
    const uint32_t nx = xs.size(), ny = ys.size(), nt = ts.size(), total = nx * ny * nt;
    float *dx, *dy, *dt, *dout;
    CHECK(hipMalloc(&dx, nx * 4));
    CHECK(hipMalloc(&dy, ny * 4));
    CHECK(hipMalloc(&dt, nt * 4));
    CHECK(hipMalloc(&dout, (size_t)total * 4));
    CHECK(hipMemcpy(dx, xs.data(), nx * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dy, ys.data(), ny * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dt, ts.data(), nt * 4, hipMemcpyHostToDevice));
    const double rho = (double)0.35f;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<float> ref(total), got(total);
    const char *names[3] = {"walk", "regroup", "regroup1"};
    for (int form = 0; form < 3; ++form)
    {
        if (!(forms >> form & 1))
            continue;
        const uint32_t blocks = form == 2 ? (total + 255) / 256 : 4096;
        std::vector<float> ms;
        for (int r = 0; r < reps + 1; ++r)
        {
            CHECK(hipMemset(dout, 0xFF, (size_t)total * 4));
            CHECK(hipEventRecord(e0, 0));
            if (form == 0)
                hipLaunchKernelGGL(k_table<0>, dim3(blocks), dim3(256), 0, 0, dx, nx, dy, ny, dt, nt, rho, dout);
            else
                hipLaunchKernelGGL(k_table<1>, dim3(blocks), dim3(256), 0, 0, dx, nx, dy, ny, dt, nt, rho, dout);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float t;
            CHECK(hipEventElapsedTime(&t, e0, e1));
            if (r)
                ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        CHECK(hipMemcpy(got.data(), dout, (size_t)total * 4, hipMemcpyDeviceToHost));
        long diff = 0;
        if (form == 0)
            ref = got;
        else if (forms & 1)
            for (uint32_t e = 0; e < total; ++e)
                diff += std::memcmp(&ref[e], &got[e], 4) != 0;
        printf("%-9s entries %u blocks %u: %.3f ms (median of %d), entries that differ from walk: %ld\n", names[form], total, blocks, ms[ms.size() / 2],
               reps, diff);
        fflush(stdout);
    }
    return 0;
}
