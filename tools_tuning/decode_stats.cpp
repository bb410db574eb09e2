#include "juliet_hip.h"
#include "decode.hpp"
using namespace jlhost;
#include <chrono>
#include <cstdio>
#include <sys/resource.h>
int main(int argc, char **argv) {
    for (int i = 0; i < 3; ++i) {
        rusage r0; getrusage(RUSAGE_SELF, &r0);
        auto t0 = std::chrono::steady_clock::now();
        IngestOptions io; if (argc > 2) io.min_qv = (uint32_t)atoi(argv[2]); size_t n = 0; RecordSink sink;
        sink.give = [&](RecordArrays &c) { n += c.pos.size(); c.clear(); };
        std::vector<BamRef> refs; std::string text;
        PipelinedBamReader::run(argv[1], io, -1, io.min_qv != 0, sink, &refs, &text, 0);
        auto t1 = std::chrono::steady_clock::now();
        rusage r1; getrusage(RUSAGE_SELF, &r1);
        printf("%zu records %.1f ms wall, minflt %ld, nvcsw %ld nivcsw %ld\n", n, std::chrono::duration<double, std::milli>(t1 - t0).count(), r1.ru_minflt - r0.ru_minflt, r1.ru_nvcsw - r0.ru_nvcsw, r1.ru_nivcsw - r0.ru_nivcsw);
    }
}
