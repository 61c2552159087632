/* TEST INFRASTRUCTURE.  The reference's test runner (test/case_main.h:171-228) selects cases by SUBSTRING of argv[1]; "cudnn forward convolution in
 * half precision" therefore also selects its "... with palettize" sibling, which aborts in the host's backend lookup (palettized weights are out of
 * scope) and takes the process down.  The reference's test sources are compiled with -Dstrstr=nnc_case_strstr (oracle/build_ref_host.sh): when
 * NNC_CASE_EXACT names the needle, the match is an exact string comparison; every other call is the C library's strstr. */
#include <stdlib.h>
#undef strstr
char* strstr(const char*, const char*);
int strcmp(const char*, const char*);
char* nnc_case_strstr(const char* hay, const char* needle)
{
	const char* const exact = getenv("NNC_CASE_EXACT");
	if (exact && strcmp(exact, needle) == 0)
		return strcmp(hay, needle) == 0 ? (char*)hay : 0;
	return strstr(hay, needle);
}
