// conv_main.cpp — the reference's CLI surface (main.go:578-645) over the MI355X engine:
//     conv <ker_wid 3|5|7> <i_batch 0..3> <num_tests <= 10>
// prints the same line shapes as the reference's `conv` run (SURVEY.md 8(a)-S "CLI output contract").
// It runs the slot-packed "Base Line" (hconv_bl.cpp, scope row 8f-2) and then "Ours" (hconv_host.cpp), as main.go:639-643 does.
// `convReLU k i n` runs both with their bootstrapping chains (hconv_relu.cpp, scope row 8f-1: the baseline's stock Bootstrapp over
// parameter set [7] and Ours' CtoS / StoC over set [6]); `resnet ker depth 1 n false` runs the encrypted ResNet inference (hconv_resnet.cpp, scope row 8f-3).
// HCONV_SKIP_BL=1 skips the baseline half (not a reference feature; for timing "Ours" alone).
// `conv --test-mode <args>` honours the test-only overrides HCONV_SEED / HCONV_CHAIN_REPLAY; without the flag they are fatal when set.
#include <stdio.h>
#include <stdlib.h>
#include <string>

#include "hconv_host.hpp"

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "--test-mode") {      // not a reference feature: enables HCONV_SEED / HCONV_CHAIN_REPLAY (hconv_host.hpp)
        hconv::testMode() = true; argv[1] = argv[0]; argv++; argc--;
        fprintf(stderr, "hconv: --test-mode: test-only overrides from the environment are honoured\n");
    }
    const int batchs[5] = {4, 16, 64, 256, 1024}, widths[5] = {128, 64, 32, 16, 8};   // main.go:578-579
    if (argc < 5) hconv::panic("runtime error: index out of range (usage: conv|convReLU <ker_wid> <i_batch> <num_tests>)");
    const std::string test_name = argv[1];
    // The bootstrapping chains allocate and free a few buffers per evaluator operation, and hipFree drains the device every time: the chain commands run on cached
    // allocations (hconv.hip hcx_malloc) unless HCONV_ASYNC_ALLOC=0 says otherwise - ResNet-20 at 8 images per launch set 2.12 -> 2.02 s per set (profiles/round4_cached_alloc_ab.txt)
    if (test_name == "convReLU" || test_name == "resnet") setenv("HCONV_ASYNC_ALLOC", "1", 0);
    const int ker_wid = atoi(argv[2]), i_batch = atoi(argv[3]), num_tests = atoi(argv[4]);
    if (!(ker_wid == 3 || ker_wid == 5 || ker_wid == 7)) hconv::panic("Wrong kernel wid (not in 3,5,7)");
    bool boot = false;
    if (test_name == "conv") {
        if (num_tests > 10 || i_batch > 3) hconv::panic("Too many tests (>10) or too many batch index (>3)");
    } else if (test_name == "convReLU") {
        boot = true;
        if (num_tests > 10 || i_batch > 3) hconv::panic("Too many tests (>10) or too many batch index (>3)");
    } else if (test_name == "resnet") {                                   // main.go:609-621: resnet ker depth wide_case test_num cf100
        if (argc < 7) hconv::panic("runtime error: index out of range (usage: resnet <ker_wid> <depth> <wide_case> <test_num> <cf100>)");
        const int depth = atoi(argv[3]), wide_case = atoi(argv[4]), test_num = atoi(argv[5]);
        const bool cf100 = std::string(argv[6]) == "true" || std::string(argv[6]) == "1";
        if (wide_case < 1 || wide_case > 3) hconv::panic("Wrong wide case!");                              // main.go:628
        // BASELINE config 5 is `resnet 3 20 1 n false`; the wide networks and the CIFAR-100 head are outside the scope table and not built
        if (wide_case != 1 || cf100) hconv::panic("resnet: wide_case 2 / 3 and cf100 = true are out of scope in this build (SURVEY.md section 2, row 14)");
        hconv::testResNet_crop_sparse(0, test_num, ker_wid, depth, false);
        return 0;
    } else hconv::panic("wrong test type");
    if (i_batch < 0) hconv::panic("runtime error: index out of range");
    if (boot) printf("Convolution followed by ReLU (& Bootstrapping) test start!\n");
    else printf("Convolution test start! (No Bootstrapping)\n");
    printf("Ker:  %d batches:  %d widths:  %d\n", ker_wid, batchs[i_batch], widths[i_batch]);
    printf("Base Line start.\n");
    if (getenv("HCONV_SKIP_BL") && atoi(getenv("HCONV_SKIP_BL"))) printf("(HCONV_SKIP_BL set: baseline skipped)\n");
    else hconv::testConv_BL_in(batchs[i_batch], widths[i_batch], ker_wid, num_tests, boot);
    printf("Ours start.\n");
    hconv::testConv_in(batchs[i_batch], widths[i_batch], ker_wid, num_tests, boot);
    return 0;
}
