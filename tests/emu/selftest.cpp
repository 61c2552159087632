// Emulator self-test: MFMA lane maps + barriers + shuffles.
#include <hip/hip_runtime.h>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k_mfma(const float* A, const float* B, float* C, int K)
{ // one wave: C[32][32] = A[32][K] * B[K][32]
	const int l = threadIdx.x, i = l & 31, h = l >> 5;
	floatx16 acc = {0};
	for (int k = 0; k < K; k += 2)
		acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + h], B[(k + h) * 32 + i], acc, 0, 0, 0);
	for (int r = 0; r < 16; r++) { const int row = (r & 3) + 8 * (r >> 2) + 4 * h; C[row * 32 + i] = acc[r]; }
}
__global__ void k_reduce(const float* x, float* out, int n)
{
	__shared__ float part[4];
	float s = 0;
	for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
	for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
int main()
{
	const int K = 8;
	std::vector<float> A(32 * K), B(K * 32), C(32 * 32), R(32 * 32, 0);
	for (int i = 0; i < 32 * K; i++) { A[i] = (i * 7 % 13) - 6; B[i] = (i * 5 % 11) - 5; }
	hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, A.data(), B.data(), C.data(), K);
	for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) for (int k = 0; k < K; k++) R[i * 32 + j] += A[i * K + k] * B[k * 32 + j];
	for (int i = 0; i < 1024; i++) if (C[i] != R[i]) { printf("MFMA mismatch at %d: %f vs %f\n", i, C[i], R[i]); return 1; }
	std::vector<float> x(1000, 1.0f), o(3);
	hipLaunchKernelGGL(k_reduce, dim3(3), dim3(256), 0, 0, x.data(), o.data(), 1000);
	if (o[0] != 1000 || o[2] != 1000) { printf("reduce mismatch %f\n", o[0]); return 1; }
	printf("emu selftest OK\n");
	return 0;
}
