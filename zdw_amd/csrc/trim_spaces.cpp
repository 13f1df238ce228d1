// trim_spaces — the helper `convertDWfile -v -t` pipes the source file through before comparing it with the decoded file
// (cplusplus/ConvertToZDW.cpp:205-211 names it; the reference repository does not ship it).  Writes the TSV file to stdout
// with the trailing blanks of every field removed, as the encoder's -t does before it looks at a field
// (cplusplus/ConvertToZDW.cpp:295-313): a field ends at a tab or newline that an even number of backslashes precedes.
#include <cstdio>
#include <string>

int main(int argc, char **argv)
{
    if (argc != 2) { fprintf(stderr, "usage: %s <file.sql>\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    std::string field;
    size_t backslashes = 0;           // length of the run of backslashes that ends at the last byte read
    auto flush = [&](int sep) {
        size_t n = field.size();
        while (n && field[n - 1] == ' ') --n;
        fwrite(field.data(), 1, n, stdout);
        if (sep >= 0) fputc(sep, stdout);
        field.clear();
    };
    for (int c; (c = fgetc(f)) != EOF;) {
        if ((c == '\t' || c == '\n') && backslashes % 2 == 0) { flush(c); backslashes = 0; continue; }
        field.push_back(static_cast<char>(c));
        backslashes = c == '\\' ? backslashes + 1 : 0;
    }
    if (!field.empty()) flush(-1);
    fclose(f);
    return ferror(stdout) ? 1 : 0;
}
