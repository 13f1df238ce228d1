// convertDWfile — command line of the MI355X ZDW encoder.  Same argv grammar, messages and exit
// codes as the reference tool (cplusplus/convertDWfile.cpp:36-67,98-260): flags are parsed in one
// pass, then every non-flag argument is converted in order; any per-file failure turns the exit
// code into CONVERSION_FAILED (2) while the detailed code is printed to stderr.
#include "ConvertToZDW.h"
#include "parallel_gzip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unistd.h>

using adobe::zdw::ConvertToZDW;

namespace {

void showVersion()
{
    printf("ConvertToZDW, Version %i%s\n", ConvertToZDW::CONVERT_ZDW_CURRENT_VERSION, ConvertToZDW::CONVERT_ZDW_VERSION_TAIL);
}

void usage(const char *executable)
{
    const char *exe = strrchr(executable, '/');
    exe = exe ? exe + 1 : executable;
    printf("Usage: %s [-d <dir>] [-(b|J|Jf|z|q|r|v)] [other options] file1 [file2] ...\n", exe);
    fputs("\t-b  compress .zdw with bzip2 [default=use gzip]\n"
          "\t-J  compress .zdw with xz [default=use gzip]\n"
          "\t -Jf  compress to .zdw.xz file via fsx (applying fastlzma2 algorithm)\n"
          "\t-z  compress .zdw with zstd\n"
          "\t-d  output to directory <dir> [default=same directory as source file]\n"
          "\t-i  streaming input from stdin; file1 is used as the implied name for the input stream\n"
          "\t-q  quiet operation (no status or progress messages) [default=not quiet]\n"
          "\t-r  remove the old files\n"
          "\t-t  trim trailing spaces from fields (for MySQL 5 exports)\n"
          "\t-v  validate the new file\n"
          "\n"
          "\t--zargs=X          arguments to pass in to the file compression process\n"
          "\t--mem-limit=<MB>   limit the MB of RAM used (default=3072 MB)\n"
          "\n"
          "\t--metadata:<key>=<value>   supply a key-value pair to store as file metadata for every file being converted\n"
          "\t--metadata-file=<filename> supply a filepath to specify key-value pairs (formatted as '<key>=<value>' pairs, each on a separate line) to store as file metadata for every file being converted\n"
          "\n"
          "\t--help     show this help\n"
          "\t--version  show the version number\n"
          "\n"
          "\t--format-version=<9|10|11>  (this build) write an older format version: 10 has no metadata block, 9 also no mediumtext/longtext\n"
          "Input files must have a .sql extension.\n"
          "\n",
          stdout);
}

ConvertToZDW::ERR_CODE reportError(ConvertToZDW::ERR_CODE res)
{
    int idx = res;
    if (idx >= ConvertToZDW::ERR_CODE_COUNT) idx = ConvertToZDW::UNKNOWN_ERROR;
    fprintf(stderr, "ZDW conversion failed.  Internal error code=%i (%s)\n", res, ConvertToZDW::ERR_CODE_TEXTS[idx]);
    return res;
}

int badParam(const char *exe, const char *param)
{
    fprintf(stderr, "%s: Unknown parameter '%s'\n\n", exe, param);
    fprintf(stderr, "    Run with --help for usage info.\n");
    return ConvertToZDW::BAD_PARAMETER;
}

}  // namespace

// The whole command line as a callable, so that tests can drive it in-process (one HIP
// initialisation for hundreds of cases); `main` below is a one-line wrapper.
static int runCommandLine(int argc, char *argv[])
{
    const char *program = argv[0];
    if (argc < 2) {
        showVersion();
        usage(program);
        return ConvertToZDW::NO_ARGS;
    }

    bool streaming = false, removeOld = false, trim = false, validate = false, quiet = false;
    double memLimitMB = 3072.0;            // DEFAULT_PROCESS_MEMORY_THRESHOLD of the reference (memory.cpp:22)
    std::vector<int> devices;
    ConvertToZDW::Compressor compressor = ConvertToZDW::GZIP;
    const char *outputDir = NULL, *zArgs = NULL;
    std::map<std::string, std::string> metadata;
    int files = 0, formatVersion = ConvertToZDW::CONVERT_ZDW_CURRENT_VERSION;

    for (int i = 1; i < argc; ++i) {
        const char *a = argv[i];
        if (a[0] != '-') {
            if (streaming && files > 0) return reportError(ConvertToZDW::TOO_MANY_INPUT_FILES);
            ++files;
            continue;
        }
        switch (a[1]) {
        case 'b': compressor = ConvertToZDW::BZIP2; break;
        case 'J': compressor = a[2] == 'f' ? ConvertToZDW::FXZ : ConvertToZDW::XZ; break;
        case 'z': compressor = ConvertToZDW::ZSTD; break;
        case 'd':
            if (++i >= argc) { usage(program); return ConvertToZDW::MISSING_ARGUMENT; }
            outputDir = argv[i];
            break;
        case 'i': streaming = true; break;
        case 'q': quiet = true; break;
        case 'r': removeOld = true; break;
        case 't': trim = true; break;
        case 'v': validate = true; break;
        case '-': {
            const char *flag = a + 2;
            if (!strcmp(flag, "help")) { showVersion(); usage(program); return ConvertToZDW::OK; }
            if (!strcmp(flag, "ver") || !strcmp(flag, "version")) { showVersion(); return ConvertToZDW::OK; }
            if (!strncmp(flag, "mem-limit=", 10)) {
                // The reference ends a block when its process is at or beyond this many MB of virtual size while the string
                // heap takes a new arena (memory.cpp:60-81, stringheap.cpp:74-86).  Here that size is a deterministic model
                // (zdw_opts.mem_limit_mb): same flag, same default, block ends where the reference build's would.
                if (!(atof(flag + 10) > 0.0)) return badParam(program, a);
                memLimitMB = atof(flag + 10);
                break;
            }
            if (!strncmp(flag, "metadata:", 9)) {
                const char *key = flag + 9, *eq = strchr(key, '=');
                if (!eq) return badParam(program, a);
                metadata[std::string(key, eq - key)] = std::string(eq + 1);
                break;
            }
            if (!strncmp(flag, "metadata-file=", 14)) {
                const char *path = flag + 14;
                const int bad = ConvertToZDW::loadMetadataFile(path, metadata);
                if (bad) {
                    fprintf(stderr, "%s: Metadata file load error '%s' (line %d)\n\n", program, path, bad);
                    return ConvertToZDW::BAD_PARAMETER;
                }
                break;
            }
            if (!strncmp(flag, "zargs=", 6)) { zArgs = flag + 6; break; }
            if (!strncmp(flag, "gpus=", 5)) {
                // not in the reference tool: row-shard the file's block over the first N devices of the node
                const int ngpu = atoi(flag + 5);
                if (ngpu < 1 || ngpu > 64) return badParam(program, a);
                devices.clear();
                for (int d = 0; d < ngpu; ++d) devices.push_back(d);
                break;
            }
            if (!strncmp(flag, "devices=", 8)) {
                // ... or over exactly these devices (a device may be listed more than once)
                devices.clear();
                for (const char *q = flag + 8; *q;) {
                    char *end = NULL;
                    const long d = strtol(q, &end, 10);
                    if (end == q || d < 0) return badParam(program, a);
                    devices.push_back(static_cast<int>(d));
                    q = *end == ',' ? end + 1 : end;
                    if (*end && *end != ',') return badParam(program, a);
                }
                if (devices.empty()) return badParam(program, a);
                break;
            }
            if (!strncmp(flag, "format-version=", 15)) {
                // not in the reference tool: write an older header for readers that stop at version 10 (or 9)
                char *end = NULL;
                const long ver = strtol(flag + 15, &end, 10);
                if (end == flag + 15 || *end || ver < 9 || ver > ConvertToZDW::CONVERT_ZDW_CURRENT_VERSION) return badParam(program, a);
                formatVersion = (int)ver;
                break;
            }
            usage(program);
            return badParam(program, a);
        }
        default: return badParam(program, a);
        }
    }
    if (files == 0) return reportError(ConvertToZDW::NO_INPUT_FILES);
    if (streaming && isatty(0)) return reportError(ConvertToZDW::NO_INPUT_FILES);

    ConvertToZDW::ERR_CODE exitCode = ConvertToZDW::OK;
    for (int i = 1; i < argc; ++i) {
        if (argv[i][0] == '-') {
            if (argv[i][1] == 'd') ++i;   // skip the directory argument
            continue;
        }
        char filestub[1024];
        ConvertToZDW convert(quiet, streaming);
        convert.setMemoryLimitMB(memLimitMB);
        if (!devices.empty()) convert.setDevices(devices);
        convert.setFormatVersion(formatVersion);
        convert.compressor = compressor;
        if (trim) convert.trimTrailingSpaces();
        const ConvertToZDW::ERR_CODE res = convert.convertFile(argv[i], program, validate, filestub, outputDir, zArgs, metadata);
        if (res != ConvertToZDW::OK) {
            if (!quiet) reportError(res);
            exitCode = ConvertToZDW::CONVERSION_FAILED;
        }
        if (removeOld) {
            if (res != ConvertToZDW::OK) {
                fprintf(stderr, "Could not remove original %s file because conversion was not good\n", filestub);
            } else {
                const std::string stub(filestub);
                unlink((stub + ".desc." + convert.getInputFileExtension()).c_str());
                unlink((stub + "." + convert.getInputFileExtension()).c_str());
            }
        }
    }
    return exitCode;
}

// The writer behind `.zdw.gz` on its own (parallel_gzip.h): `n` bytes handed over in pieces of `piece` bytes, deflated on `threads`
// threads into `path`.  0: written.  (tests: the file must read back as the bytes, with gzip -d and with zlib)
extern "C" int zdw_cli_gzip_bytes(const unsigned char *data, unsigned long long n, const char *path, unsigned threads, int level, unsigned long long piece)
{
    FILE *f = fopen(path, "wb");
    if (!f) return 2;
    int rc = 0;
    {
        zdwhost::ParallelGzip gz(f, threads, level);
        if (!piece) piece = n ? n : 1;
        for (unsigned long long o = 0; o < n && rc == 0; o += piece)
            if (!gz.write(data + o, (size_t)(n - o < piece ? n - o : piece))) rc = 1;
        if (!gz.close()) rc = 1;
    }
    if (fclose(f) != 0) rc = 1;
    return rc;
}

extern "C" int zdw_convertDWfile_main(int argc, char *argv[])
{
    const int rc = runCommandLine(argc, argv);
    fflush(stdout);
    fflush(stderr);
    return rc;
}

#ifndef ZDW_CLI_AS_LIBRARY
int main(int argc, char *argv[]) { return zdw_convertDWfile_main(argc, argv); }
#endif
