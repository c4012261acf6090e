// Small HIP host helpers: error -> exception, RAII device buffer.
#ifndef SPA_HIP_UTIL_HPP
#define SPA_HIP_UTIL_HPP
#include <hip/hip_runtime_api.h>
#include <stdexcept>
#include <string>
#include <cstddef>
#include <cstdlib>
#include <atomic>

namespace spa {

struct HipError :public std::runtime_error
{
	explicit HipError( const std::string& msg) :std::runtime_error( msg){}
};

#define HIP_CHECK( EXPR) do { hipError_t e_ = (EXPR); if (e_ != hipSuccess) \
	throw spa::HipError( std::string("HIP error: ") + hipGetErrorString( e_) + " in " #EXPR); } while (0)

inline std::atomic<unsigned long long>& allocFailureThreshold() { static std::atomic<unsigned long long> v( 0); return v; }

struct DeviceBuffer
{
	void* ptr;
	size_t bytes;
	DeviceBuffer() :ptr(0),bytes(0){}
	~DeviceBuffer() { release(); }
	void release() { if (ptr) { (void)hipFree( ptr); ptr = 0; bytes = 0; } }
	// exact (re)allocation
	void alloc( size_t n)
	{
		release();
		if (n == 0) n = 16;
		// test hook (sp_test_fail_alloc_above, tests/test_l2_gpu.py): allocations of at least this many bytes fail like an
		// out-of-memory hipMalloc; 0 = off.  Set through the C-ABI only: nothing on the allocation path reads the environment.
		const unsigned long long failAbove = allocFailureThreshold().load( std::memory_order_relaxed);
		if (failAbove && n >= failAbove) throw HipError( "HIP error: out of memory (injected by sp_test_fail_alloc_above) in hipMalloc");
		HIP_CHECK( hipMalloc( &ptr, n));
		bytes = n;
	}
	// grow-only
	void reserve( size_t n)
	{
		if (n > bytes) alloc( n + n/4);
	}
	void upload( const void* src, size_t n)
	{
		alloc( n);
		if (n) HIP_CHECK( hipMemcpy( ptr, src, n, hipMemcpyHostToDevice));
	}
private:
	DeviceBuffer( const DeviceBuffer&);
	void operator=( const DeviceBuffer&);
};

} // namespace
#endif
