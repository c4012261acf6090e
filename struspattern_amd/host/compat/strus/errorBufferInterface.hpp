// Stand-in for strusBase's strus/errorBufferInterface.hpp (not available in this build
// environment).  Only what the pattern module uses (src/errorUtils.hpp:22-140 of strusPattern and
// its tests).  Drop this directory from the include path when the real strus headers are present.
#ifndef _STRUS_ERROR_BUFFER_INTERFACE_HPP_INCLUDED
#define _STRUS_ERROR_BUFFER_INTERFACE_HPP_INCLUDED
#include <cstdarg>
namespace strus {
enum ErrorCode {ErrorCodeUnknown=0, ErrorCodeOutOfMem, ErrorCodeRuntimeError, ErrorCodeLogicError, ErrorCodeUncaughtException,
	ErrorCodeInvalidArgument, ErrorCodeSyntax, ErrorCodeNotImplemented};
class DebugTraceInterface;
class ErrorBufferInterface
{
public:
	virtual ~ErrorBufferInterface(){}
	virtual void report( int errorcode, const char* format, ...)=0;
	virtual bool hasError() const=0;
	virtual const char* fetchError()=0;
	virtual DebugTraceInterface* debugTrace() const {return 0;}
};
}
#endif
