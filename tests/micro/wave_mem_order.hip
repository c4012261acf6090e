// Micro-experiment: is a global store by some lanes of a wave visible to a following global load
// (same address) issued by all lanes of the same wave, without an explicit wait?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define LANE (threadIdx.x & 63u)
__global__ void k_partial_store( unsigned* buf, const unsigned* idx, unsigned n, unsigned* errs)
{
	unsigned* my = buf + (blockIdx.x*4 + (threadIdx.x>>6))*256;
	unsigned bad = 0;
	for (unsigned it=1; it<=n; ++it)
	{
		unsigned h = idx[ it & 15];		// uniform, opaque to the compiler
		if (LANE < 16) my[ LANE] = it;		// partial-wave store
		unsigned v = my[ h];			// all lanes load one address
		if (v != it) bad++;
		my[ 32 + h] = v;			// all lanes store same value same address
		unsigned v2 = my[ 32 + idx[ (it+1)&15] - 1 + 1 - (idx[(it+1)&15]-h)];	// == my[32+h]
		if (v2 != v) bad++;
	}
	if (bad) atomicAdd( errs, bad);
}
int main()
{
	unsigned *buf, *idx, *errs; unsigned hidx[16]; for (int i=0;i<16;++i) hidx[i] = (i*7)&15;
	hipMalloc( &buf, 1024*4*256*4); hipMalloc( &idx, 64); hipMalloc( &errs, 4);
	hipMemcpy( idx, hidx, 64, hipMemcpyHostToDevice); hipMemset( errs, 0, 4); hipMemset( buf, 0xff, 1024*4*256*4);
	for (int blocks : {1, 64, 1024})
	{
		hipMemset( errs, 0, 4);
		hipLaunchKernelGGL( k_partial_store, dim3(blocks), dim3(256), 0, 0, buf, idx, 100000u, errs);
		hipError_t e = hipDeviceSynchronize();
		unsigned h=0; hipMemcpy( &h, errs, 4, hipMemcpyDeviceToHost);
		printf( "blocks=%d err=%s mismatches=%u\n", blocks, hipGetErrorString(e), h);
	}
	return 0;
}
