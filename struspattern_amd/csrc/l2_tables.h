// Level-2 device table records shared by the host compiler (l2_compile.cpp) and the HIP kernel
// (l2_kernel.hip).  Plain PODs of u32 words, every record a multiple of 16 bytes.
#ifndef SPA_L2_TABLES_H
#define SPA_L2_TABLES_H
#include <stdint.h>

#if defined(__HIPCC__)
#define SPA_HD __host__ __device__
#else
#define SPA_HD
#endif

namespace spa {

// signal types of a trigger (numbering of the reference, src/ruleMatcherAutomaton.hpp:48)
enum {SIG_ANY=0, SIG_SEQUENCE=1, SIG_SEQUENCE_IMM=2, SIG_WITHIN=3, SIG_DEL=4, SIG_AND=5};

// ---- device table records (all u32 words, 16-byte multiples) ----
struct DevProgram		// 32 B
{
	uint32_t initsigval;
	uint32_t initcount;
	uint32_t event;		// follow event emitted on match (0 = none)
	uint32_t resultHandle;	// result emitted on match (0 = none)
	uint32_t formatHandle;
	uint32_t positionRange;
	uint32_t trigBegin;	// first trigger template in trigdefs[]
	uint32_t trigCount;
};
struct DevTrigDef		// 16 B, stored in installation order (= last expression argument first)
{
	uint32_t event;
	uint32_t sigval;
	uint32_t variable;
	uint32_t flags;		// bits 0..3 sigtype, bit 8 isKeyEvent
};
struct DevKeyEntry		// 16 B, open addressing (linear probing), event==0 = empty
{
	uint32_t event;
	uint32_t listBegin;	// programs keyed by this event: keylist[listBegin .. listBegin+listCount)
	uint32_t listCount;
	uint32_t stopIdx;	// 1-based slot in the per-document stop-word log, 0 = not a stop word
};
struct DevKeyRef		// 16 B
{
	uint32_t program;	// 0-based index into programs[]
	uint32_t pastEvent;	// original key event of an alt-keyed program (0 = none)
	uint32_t pastStopIdx;	// its stop-word log slot
	uint32_t _pad;
};


// hash of the key-event table (open addressing, linear probing)
SPA_HD static inline uint32_t keyHash( uint32_t a)
{
	a ^= a >> 16; a *= 0x7feb352dU;
	a ^= a >> 15; a *= 0x846ca68bU;
	a ^= a >> 16;
	return a;
}

} // namespace
#endif
