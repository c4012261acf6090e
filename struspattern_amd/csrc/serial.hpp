// Byte-stream writer / reader for the compiled-table blobs ("an on-disk / wire format for the
// NFA tables and the ProgramTable so that many processes load instead of recompiling"; the reference has no
// serialisation, so the format is this library's own).  Little endian, every blob = magic, version, payload,
// FNV-1a 64 checksum of everything before it.
#ifndef SPA_SERIAL_HPP
#define SPA_SERIAL_HPP
#include <stdint.h>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace spa {

inline uint64_t blobChecksum( const uint8_t* p, size_t n)
{
	uint64_t h = 1469598103934665603ull;
	for (size_t i=0; i<n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
	return h;
}

class BlobWriter
{
public:
	explicit BlobWriter( const char* magic8) { m_buf.insert( m_buf.end(), magic8, magic8+8); }
	void u32( uint32_t v) { raw( &v, 4); }
	void u64( uint64_t v) { raw( &v, 8); }
	void f32( float v) { raw( &v, 4); }
	void f64( double v) { raw( &v, 8); }
	void str( const std::string& s) { u32( (uint32_t)s.size()); raw( s.data(), s.size()); }
	template <class T> void vec( const std::vector<T>& v) { u64( v.size()); if (!v.empty()) raw( v.data(), v.size()*sizeof(T)); }
	void raw( const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; m_buf.insert( m_buf.end(), b, b+n); }
	std::vector<uint8_t>& finish() { const uint64_t c = blobChecksum( m_buf.data(), m_buf.size()); u64( c); return m_buf; }
private:
	std::vector<uint8_t> m_buf;
};

class BlobReader
{
public:
	BlobReader( const void* p, size_t n, const char* magic8) :m_p((const uint8_t*)p),m_n(n),m_at(8)
	{
		if (n < 16 || std::memcmp( p, magic8, 8) != 0) throw std::runtime_error( "not a compiled table blob of this library (bad magic or version)");
		uint64_t c; std::memcpy( &c, m_p + n - 8, 8);
		if (c != blobChecksum( m_p, n-8)) throw std::runtime_error( "compiled table blob is corrupt (checksum mismatch)");
		m_n = n-8;
	}
	uint32_t u32() { uint32_t v; raw( &v, 4); return v; }
	uint64_t u64() { uint64_t v; raw( &v, 8); return v; }
	float f32() { float v; raw( &v, 4); return v; }
	double f64() { double v; raw( &v, 8); return v; }
	std::string str() { const uint32_t n = u32(); need( n); std::string s( (const char*)m_p + m_at, n); m_at += n; return s; }
	template <class T> void vec( std::vector<T>& v)
	{
		const uint64_t n = u64();
		if (n > (m_n - m_at) / sizeof(T)) throw std::runtime_error( "compiled table blob is truncated");
		v.resize( (size_t)n);
		if (n) raw( v.data(), (size_t)n*sizeof(T));
	}
	void raw( void* p, size_t n) { need( n); std::memcpy( p, m_p + m_at, n); m_at += n; }
	bool atEnd() const { return m_at == m_n; }
private:
	void need( size_t n) const { if (n > m_n - m_at) throw std::runtime_error( "compiled table blob is truncated"); }
	const uint8_t* m_p; size_t m_n, m_at;
};

} // namespace
#endif
