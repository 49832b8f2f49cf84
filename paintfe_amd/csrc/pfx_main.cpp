// pfx_main.cpp — the `pfx` batch tool: argv straight into the library's CLI driver (ref: src/main.rs:180-191, src/cli.rs).
#include "../../include/pfx.h"
int main(int argc, char** argv) { return pfx_cli_main(argc, argv); }
