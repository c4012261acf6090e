/*
 * strus_pattern_amd.h -- C-ABI of the MI355X-native two-level pattern engine.
 *
 * This is the drop-in boundary underneath strus's PatternLexerInterface / PatternMatcherInterface
 * (the C++ shim in struspattern_amd/host/ implements those virtual interfaces on top of these
 * entry points; the module load point stays `modstrus_analyzer_pattern`).
 * Plain C: opaque handles, plain pointers and sizes, no C++ or torch types, no exceptions.
 * Every call returns 0 on success or a negative SP_ERR_* code; the message is retrievable with
 * sp_*_last_error(handle).  Buffers returned through `**out` parameters are owned by the caller
 * and released with sp_free().  Each entry point cites the reference interface it replaces
 * (paths relative to the strusPattern source tree).
 */
#ifndef STRUS_PATTERN_AMD_H
#define STRUS_PATTERN_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_OK                    0
#define SP_ERR_INVALID          -1   /* bad argument / call order (reference: std::runtime_error -> ErrorCodeRuntimeError) */
#define SP_ERR_NOMEM            -2   /* reference: std::bad_alloc -> ErrorCodeOutOfMem */
#define SP_ERR_DEVICE           -3   /* HIP runtime error, or no GPU / HIP extension unusable */
#define SP_ERR_COMPILE          -4   /* rule / regex compilation failed */
#define SP_ERR_MATCH            -5   /* at least one document failed, see per-document status */

/* per-document status codes written by the kernels (0 = ok) */
#define SP_DOC_OK                0
#define SP_DOC_ERR_ORDER         1   /* lexems not in ascending ordpos   (src/patternMatcher.cpp:136-139) */
#define SP_DOC_ERR_ARENA         2   /* per-document working set exceeded the arena (retry with a larger arena) */
#define SP_DOC_ERR_KEYTRIGGERS   3   /* >32 identical key events in one pattern (src/ruleMatcherAutomaton.cpp:1213-1216) */
#define SP_DOC_ERR_PASTFOLLOW    4   /* "encountered past trigger with follow" (src/ruleMatcherAutomaton.cpp:1328-1332) */
#define SP_DOC_ERR_RANGE         5   /* origsize/origpos out of range    (src/patternMatcher.cpp:144-155) */
#define SP_DOC_ERR_DATAREF       6   /* "illegal free of event data reference" (src/ruleMatcherAutomaton.cpp:726-731) */
#define SP_DOC_ERR_LEXEMSIZE     7   /* matched lexem >= 65535 bytes    (src/patternLexer.cpp:727-730) */
#define SP_DOC_ERR_INTERNAL      8   /* a list walk exceeded its bound: corrupted per-document state (bug) */
#define SP_DOC_ERR_OUTPUT        9   /* result/item output buffer too small (host entry points grow it and rerun) */

/* strus::PatternMatcherInstanceInterface::JoinOperation, in the order of the switch at
 * src/patternMatcher.cpp:401-440 */
enum sp_join_op {
	SP_OP_SEQUENCE = 0, SP_OP_SEQUENCE_IMM = 1, SP_OP_SEQUENCE_STRUCT = 2,
	SP_OP_WITHIN = 3, SP_OP_WITHIN_STRUCT = 4, SP_OP_ANY = 5, SP_OP_AND = 6
};

/* strus::analyzer::PositionBind (src/patternLexer.cpp:902-915) */
enum sp_position_bind {
	SP_BIND_CONTENT = 0, SP_BIND_SUCCESSOR = 1, SP_BIND_PREDECESSOR = 2, SP_BIND_UNIQUE = 3
};

/* analyzer::PatternLexem(id, ordpos, Position(seg, ofs), origsize)  (src/patternLexer.cpp:908).
 * 16 bytes; the segment index travels in a parallel optional array (always 0 out of the lexer). */
typedef struct sp_lexem {
	uint32_t id;
	uint32_t ordpos;
	uint32_t origpos;
	uint32_t origsize;
} sp_lexem_t;

/* analyzer::PatternMatcherResult minus strings = strus::Result (src/ruleMatcherAutomaton.hpp:273-289),
 * 36 bytes.  `handle` maps to the pattern name with sp_matcher_pattern_name(). */
typedef struct sp_result {
	uint32_t handle;
	uint32_t ordpos;
	uint32_t ordend;
	uint32_t origseg;
	uint32_t origpos;
	uint32_t origendseg;
	uint32_t origend;
	uint32_t item_begin;   /* index of the first item of this result in the items array */
	uint32_t item_count;
} sp_result_t;

/* analyzer::PatternMatcherResultItem minus strings (src/patternMatcher.cpp:182) */
typedef struct sp_result_item {
	uint32_t variable;     /* maps to the variable name with sp_matcher_variable_name() */
	uint32_t ordpos;
	uint32_t ordend;
	uint32_t origseg;
	uint32_t origpos;
	uint32_t origendseg;
	uint32_t origend;
} sp_result_item_t;

/* analyzer::PatternMatcherStatistics items (src/patternMatcher.cpp:303-318) */
typedef struct sp_matcher_stats {
	double nofProgramsInstalled;
	double nofAltKeyProgramsInstalled;
	double nofSignalsFired;
	double nofTriggersAvgActive;
} sp_matcher_stats_t;

typedef struct sp_matcher sp_matcher_t;          /* PatternMatcherInstanceInterface */
typedef struct sp_matcher_ctx sp_matcher_ctx_t;  /* PatternMatcherContextInterface (+ batch mode) */
typedef struct sp_lexer sp_lexer_t;              /* PatternLexerInstanceInterface */
typedef struct sp_lexer_ctx sp_lexer_ctx_t;      /* PatternLexerContextInterface (+ batch mode) */

/* ---- library ---- */
const char* sp_version(void);
/* number of usable HIP devices (0 when no GPU is visible); never initialises more than the runtime */
int sp_device_count(void);
void sp_free(void* p);
/* test hook (tests only): device allocations of at least `bytes` fail like an out-of-memory hipMalloc; 0 switches it off */
void sp_test_fail_alloc_above(uint64_t bytes);

/* ==== level 2: token pattern matcher ==== */

/* strus::createPatternMatcher_std + PatternMatcherInterface::createInstance
 * (src/libstrus_pattern.cpp:21-33, src/patternMatcher.cpp:718-726) */
sp_matcher_t* sp_matcher_create(void);
void sp_matcher_free(sp_matcher_t* m);
const char* sp_matcher_last_error(const sp_matcher_t* m);

/* PatternMatcherInstanceInterface (src/patternMatcher.cpp:361-671), same argument meaning */
int sp_matcher_define_term_frequency(sp_matcher_t* m, uint32_t termid, double df);                 /* :361 */
int sp_matcher_push_term(sp_matcher_t* m, uint32_t termid);                                         /* :366 */
int sp_matcher_push_expression(sp_matcher_t* m, int joinop, size_t argc, uint32_t range, uint32_t cardinality); /* :377 */
int sp_matcher_push_pattern(sp_matcher_t* m, const char* name);                                     /* :510 */
int sp_matcher_attach_variable(sp_matcher_t* m, const char* name);                                  /* :522 */
int sp_matcher_define_pattern(sp_matcher_t* m, const char* name, const char* formatstring, int visible); /* :545 */
int sp_matcher_define_option(sp_matcher_t* m, const char* name, double value);                      /* :614 */
int sp_matcher_compile(sp_matcher_t* m);                                                            /* :646 */
/* symbol tables behind result handles / item variables (patternMap / variableMap, :82-83) */
uint32_t sp_matcher_pattern_id(const sp_matcher_t* m, const char* name);
const char* sp_matcher_pattern_name(const sp_matcher_t* m, uint32_t handle);
uint32_t sp_matcher_variable_id(const sp_matcher_t* m, const char* name);
const char* sp_matcher_variable_name(const sp_matcher_t* m, uint32_t variable);
/* canonical dump of the compiled ProgramTable (test hook; format in csrc/l2_compile.hpp) */
size_t sp_matcher_dump_table(const sp_matcher_t* m, uint32_t** out);
/* which kernel a context of this matcher runs on (test / diagnostics hook): 1 = the rule set is flat (every program
 * has at most 3 triggers over input terms, position range <= 63, no rule listens to another rule's result) and its
 * documents run with their whole hot state in LDS; 0 = general kernel, `why` says what disqualifies it.
 * Both kernels produce the reference's results (src/ruleMatcherAutomaton.cpp:772-1334); the choice is not observable. */
int sp_matcher_fast_tier(const sp_matcher_t* m, char* why, size_t whysize);

/* Compiled-table serialisation (the reference has none -- every process recompiles, and with
 * Hyperscan that takes seconds for 10k patterns, src/patternLexer.cpp:1068-1118): the rule set with its key
 * index, stop words, names, format strings and options as one self-checking blob (magic, version, FNV-1a 64),
 * so that the ranks of a multi-GPU job or the workers of a service load it instead of compiling.  The blob is
 * malloc'ed (sp_free); a loaded matcher behaves like the one that was saved (same tables word for word).
 * sp_matcher_deserialize returns NULL and the reason in `err` for a blob that is truncated, corrupt or foreign. */
int sp_matcher_serialize(const sp_matcher_t* m, void** blob, size_t* size);
sp_matcher_t* sp_matcher_deserialize(const void* blob, size_t size, char* err, size_t errsize);

/* ---- result format strings (definePattern's formatstring, src/patternMatcher.cpp:561-566, :172-181, :253-262).
 * The device does not build strings: it reports which format applies and what its arguments are.
 *   - a result with a format handle: its item list holds the ARGUMENTS of the format string; the
 *     reference returns such a result with a value and an empty item list (:253-262).
 *   - an item with a format handle (a variable bound to a sub-pattern that has a format string): the
 *     `nsub` records that follow it are the arguments of ITS format (nested the same way) and are not
 *     items of the enclosing list (:172-181); items without a format are followed by their sub-items
 *     as siblings (:185-188), as in a matcher without format strings.
 * Format strings are numbered 1..sp_matcher_format_count() in definePattern order.  The formatter
 * itself (PatternResultFormat) lives in strusAnalyzer, outside the reference repository; the host
 * mirrors implement "{variable}" / "{variable|separator}" substitution as its documentation describes. */
uint32_t sp_matcher_format_count(const sp_matcher_t* m);
const char* sp_matcher_format_string(const sp_matcher_t* m, uint32_t format_handle);

/* PatternMatcherInstanceInterface::createContext (src/patternMatcher.cpp:586).  `device` is the
 * HIP device ordinal.  Fails with SP_ERR_DEVICE when no GPU is usable: there is no CPU fallback. */
sp_matcher_ctx_t* sp_matcher_ctx_create(const sp_matcher_t* m, int device);
void sp_matcher_ctx_free(sp_matcher_ctx_t* c);
const char* sp_matcher_ctx_last_error(const sp_matcher_ctx_t* c);

/* PatternMatcherContextInterface::putInput (src/patternMatcher.cpp:131): appends lexems of the
 * current document (host side, ascending ordpos); origseg may be NULL (= all 0). */
int sp_matcher_ctx_put_input(sp_matcher_ctx_t* c, const sp_lexem_t* lexems, const uint32_t* origseg, size_t n);
/* PatternMatcherContextInterface::fetchResults (:271): runs the document on the GPU. */
int sp_matcher_ctx_fetch_results(sp_matcher_ctx_t* c, sp_result_t** results, size_t* nresults,
                                 sp_result_item_t** items, size_t* nitems);
/* format handles of the results / items of the last sp_matcher_ctx_fetch_results (borrowed pointers,
 * valid until the next fetch or reset; NULL when the matcher has no format strings) */
int sp_matcher_ctx_fetch_formats(sp_matcher_ctx_t* c, const uint32_t** result_format, const uint32_t** item_format);
/* PatternMatcherContextInterface::getStatistics (:303) of the last fetch */
int sp_matcher_ctx_statistics(sp_matcher_ctx_t* c, sp_matcher_stats_t* out);
/* PatternMatcherContextInterface::reset (:320) */
int sp_matcher_ctx_reset(sp_matcher_ctx_t* c);

/* ---- batch mode: one Context per document, many documents per launch
 *      (the reference runs one Context per document per thread,
 *       tests/randomTokenPatternMatch/src/testRandomTokenPatternMatch.cpp:130-146, :325-345) ---- */
typedef struct sp_match_batch {
	size_t ndocs;
	size_t nresults;
	size_t nitems;
	sp_result_t* results;          /* grouped by document, firing order inside a document */
	sp_result_item_t* items;
	uint64_t* doc_result_offsets;  /* ndocs+1 */
	uint64_t* doc_stats;           /* ndocs x 4: programs installed, alt-key programs installed, signals fired, sum of active triggers */
	int32_t* doc_status;           /* ndocs x SP_DOC_* */
	/* matchers with format strings only (else NULL), see "result format strings" below */
	uint32_t* result_format;       /* nresults: format handle of the result (0 = none) */
	uint32_t* item_format;         /* nitems x {format handle of the item, nsub} */
} sp_match_batch_t;

/* host buffers in, host buffers out (PCIe both ways) */
int sp_matcher_ctx_match_docs(sp_matcher_ctx_t* c, const sp_lexem_t* lexems, const uint32_t* origseg,
                              const uint64_t* doc_offsets, size_t ndocs, sp_match_batch_t* out);
void sp_match_batch_free(sp_match_batch_t* b);

/* device-resident variant: d_lexems / d_origseg (may be NULL) / d_doc_offsets are device pointers,
 * the launch is asynchronous on `stream` (a hipStream_t, NULL = default stream); results stay in HBM. */
typedef struct sp_match_device_batch {
	size_t ndocs;
	void* d_results;             /* sp_result_t[result_capacity], grouped by document after sp_..._finish */
	void* d_items;               /* sp_result_item_t[] */
	void* d_doc_result_offsets;  /* uint64_t[ndocs+1] */
	void* d_doc_stats;           /* uint64_t[ndocs*4] */
	void* d_doc_status;          /* int32_t[ndocs] */
	void* d_counters;            /* uint64_t[8]: results, items, events, failed docs, ... */
	void* d_result_format;       /* uint32_t[] parallel to d_results, NULL without format strings */
	void* d_item_format;         /* uint32_t[][2] parallel to d_items, NULL without format strings */
} sp_match_device_batch_t;
int sp_matcher_ctx_match_docs_device(sp_matcher_ctx_t* c, const void* d_lexems, const void* d_origseg,
                                     const void* d_doc_offsets, size_t ndocs, size_t nlexems,
                                     void* stream, sp_match_device_batch_t* out);
/* fused pipeline entry: the lexems and per-document (first, count) ranges produced by
 * sp_lexer_ctx_match_docs_device are consumed in place, nothing leaves HBM in between */
int sp_matcher_ctx_match_lexed_device(sp_matcher_ctx_t* c, const void* d_lexems, const void* d_doc_ranges,
                                      size_t ndocs, size_t nlexems_hint, void* stream, sp_match_device_batch_t* out);
/* copies the results of the last device batch to the host, grouped by document (as sp_matcher_ctx_match_docs
 * returns them, `exclusive` elimination included) */
int sp_matcher_ctx_batch_fetch(sp_matcher_ctx_t* c, sp_match_batch_t* out);
/* the same for the documents [first_doc, first_doc+ndocs) of the last device batch only (a caller that checks a
 * sample, or picks up the documents of one shard, does not move the whole batch over PCIe) */
int sp_matcher_ctx_batch_fetch_docs(sp_matcher_ctx_t* c, size_t first_doc, size_t ndocs, sp_match_batch_t* out);
/* waits for the stream and returns counters[0..7] = {results, items, events, failed docs, documents the LDS-resident
 * kernel handed over to the general one (diagnostics), 0..} */
int sp_matcher_ctx_batch_counters(sp_matcher_ctx_t* c, uint64_t counters[8]);
/* duration of the last rule-automaton kernel in milliseconds (HIP events on the launch stream) */
double sp_matcher_ctx_last_kernel_ms(sp_matcher_ctx_t* c);
/* which kernel the context runs its batches on: 0 general automaton, 1 LDS-resident automaton (flat rule sets),
 * 2 the opt-in join prototype (environment SPA_L2_JOIN=1 at context creation; result SETS with their items, no statistics) */
int sp_matcher_ctx_kernel_kind(const sp_matcher_ctx_t* c);
/* name of the kernel that does the work of this context's batches (what rocprofv3 lists) */
const char* sp_matcher_ctx_kernel_name(const sp_matcher_ctx_t* c);
/* copies the per-document status words of the last batch to the host (waits for the stream) */
int sp_matcher_ctx_batch_status(sp_matcher_ctx_t* c, int32_t* status, size_t ndocs);
/* doubles every per-document working-set capacity (what the host entry points do on SP_DOC_ERR_ARENA) */
int sp_matcher_ctx_grow_arena(sp_matcher_ctx_t* c);
/* minimum capacity (records) of the device result / item buffers of the batch entry points */
int sp_matcher_ctx_reserve_output(sp_matcher_ctx_t* c, uint64_t results, uint64_t items);
/* working-set capacity per in-flight document; 0 keeps a default.  Takes effect at the next launch. */
int sp_matcher_ctx_set_arena(sp_matcher_ctx_t* c, uint32_t max_rules, uint32_t max_triggers, uint32_t bucket_capacity,
                             uint32_t max_items, uint32_t max_follow);

/* ==== level 1: pattern lexer ==== */

/* strus::createPatternLexer_std + PatternLexerInterface::createInstance
 * (src/libstrus_pattern.cpp:35-47, src/patternLexer.cpp:1165-1172) */
sp_lexer_t* sp_lexer_create(void);
void sp_lexer_free(sp_lexer_t* l);
const char* sp_lexer_last_error(const sp_lexer_t* l);

/* PatternLexerInstanceInterface (src/patternLexer.cpp:971-1141), same argument meaning.
 * Expressions are NUL terminated; posbind is an sp_position_bind. */
int sp_lexer_define_lexem_name(sp_lexer_t* l, uint32_t id, const char* name);                       /* :971 */
const char* sp_lexer_get_lexem_name(const sp_lexer_t* l, uint32_t id);                              /* :983 */
int sp_lexer_define_lexem(sp_lexer_t* l, uint32_t id, const char* expression, uint32_t resultIndex,
                          uint32_t level, int posbind);                                             /* :990 */
int sp_lexer_define_symbol(sp_lexer_t* l, uint32_t symbolid, uint32_t patternid, const char* name); /* :1008 */
uint32_t sp_lexer_get_symbol(const sp_lexer_t* l, uint32_t patternid, const char* name);            /* :1020 */
int sp_lexer_define_option(sp_lexer_t* l, const char* name, double value);                          /* :1031 */
int sp_lexer_compile(sp_lexer_t* l);                                                                /* :1068 */
/* compiled lexer (automaton tables, whole-word literal table, symbol tables, lexem names) as a blob and back; only a
 * compiled lexer can be saved, a loaded one is compiled and frozen (see sp_matcher_serialize) */
int sp_lexer_serialize(const sp_lexer_t* l, void** blob, size_t* size);
sp_lexer_t* sp_lexer_deserialize(const void* blob, size_t size, char* err, size_t errsize);
/* compiled automaton tables as a flat u64 array (test hook; layout in csrc/capi_l1.cpp) */
size_t sp_lexer_dump_tables(const sp_lexer_t* l, uint64_t** out);

/* PatternLexerInstanceInterface::createContext (:1120): an error before compile(); SP_ERR_DEVICE
 * without a usable GPU (no CPU fallback) */
sp_lexer_ctx_t* sp_lexer_ctx_create(const sp_lexer_t* l, int device);
void sp_lexer_ctx_free(sp_lexer_ctx_t* c);
const char* sp_lexer_ctx_last_error(const sp_lexer_ctx_t* c);
/* PatternLexerContextInterface::match (:858): one document, host buffers */
int sp_lexer_ctx_match(sp_lexer_ctx_t* c, const char* src, size_t srclen, sp_lexem_t** lexems, size_t* nlexems);
/* PatternLexerContextInterface::reset (:700) */
int sp_lexer_ctx_reset(sp_lexer_ctx_t* c);

/* ---- batch mode: many documents per launch ---- */
typedef struct sp_lex_batch {
	size_t ndocs;
	size_t nlexems;
	sp_lexem_t* lexems;            /* grouped by document, reference output order inside a document */
	uint64_t* doc_lexem_offsets;   /* ndocs+1 */
	int32_t* doc_status;           /* ndocs x SP_DOC_* */
} sp_lex_batch_t;
/* text: all documents back to back; doc_offsets: ndocs+1 byte offsets */
int sp_lexer_ctx_match_docs(sp_lexer_ctx_t* c, const char* text, const uint64_t* doc_offsets, size_t ndocs, sp_lex_batch_t* out);
void sp_lex_batch_free(sp_lex_batch_t* b);

typedef struct sp_lex_device_batch {
	size_t ndocs;
	void* d_lexems;        /* sp_lexem_t[], each document's lexems contiguous */
	void* d_doc_ranges;    /* uint64_t[ndocs][2] = (first lexem, count) */
	void* d_doc_status;    /* int32_t[ndocs] */
	void* d_counters;      /* uint64_t[8]: lexems, bytes, raw reports, failed docs */
} sp_lex_device_batch_t;
/* device-resident text and offsets, asynchronous on `stream` (hipStream_t); lexems stay in HBM */
int sp_lexer_ctx_match_docs_device(sp_lexer_ctx_t* c, const void* d_text, const void* d_doc_offsets,
                                   size_t ndocs, size_t nbytes, void* stream, sp_lex_device_batch_t* out);
/* host copy of the lexems of the documents [first_doc, first_doc+ndocs) of the last device batch
 * (what PatternLexerContextInterface::match returns for each of them, src/patternLexer.cpp:858) */
int sp_lexer_ctx_batch_fetch_docs(sp_lexer_ctx_t* c, size_t first_doc, size_t ndocs, sp_lex_batch_t* out);
int sp_lexer_ctx_batch_counters(sp_lexer_ctx_t* c, uint64_t counters[8]);
int sp_lexer_ctx_batch_status(sp_lexer_ctx_t* c, int32_t* status, size_t ndocs);
double sp_lexer_ctx_last_kernel_ms(sp_lexer_ctx_t* c);
/* the same interval split at the boundary of the lexer's two kernels (automaton scan; literals + start of match +
 * handler + ordinal positions) */
int sp_lexer_ctx_last_kernel_ms_split(sp_lexer_ctx_t* c, double* scan_ms, double* post_ms);
/* the same in three parts: automaton scan; words kernel (whole-word literals and word shapes, found where runs of word
 * characters end); start of match + handler + ordinal positions */
int sp_lexer_ctx_last_kernel_ms_split3(sp_lexer_ctx_t* c, double* scan_ms, double* words_ms, double* post_ms);
/* name of the scan kernel the last launch of this context went through ("(none)" before the first launch) */
const char* sp_lexer_ctx_scan_kernel_name(const sp_lexer_ctx_t* c);
/* the same for the words kernel (whole-word literals and word shapes): its 12- or 16-wave instance, "(none)" for tables it does not take */
const char* sp_lexer_ctx_words_kernel_name(const sp_lexer_ctx_t* c);
int sp_lexer_ctx_reserve_output(sp_lexer_ctx_t* c, uint64_t lexems);
int sp_lexer_ctx_grow_arena(sp_lexer_ctx_t* c);

#ifdef __cplusplus
}
#endif
#endif
