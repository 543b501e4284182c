# integration/reference_host.mk - shared by integration/Makefile (the PRODUCT binding: katago_hip) and oracle/Makefile (the test
# binaries bound to the CPU oracle): the reference's host sources compiled WHERE THEY LIE under $(REF) with plain g++ (the reference's
# cmake build is not used) and this repository's integration sources, all into integration/_build/obj (git-ignored; nothing but
# objects and binaries is ever written, and nothing under $(REF)). `make` needs $(REF); on the GPU box it does not exist and the
# prebuilt binaries are used.

REF      ?= /root/reference
REPO     := $(abspath $(dir $(lastword $(MAKEFILE_LIST)))/..)
BUILD    := $(REPO)/integration/_build
OBJ      := $(BUILD)/obj
CXX      ?= g++
CC       ?= gcc

REF_CXXFLAGS  := -std=c++17 -O2 -DNDEBUG -DNO_GIT_REVISION -DNO_LIBZIP -pthread -w \
                 -I$(REF)/cpp -I$(REF)/cpp/external -isystem $(REF)/cpp/external/tclap-1.2.5/include \
                 -isystem $(REF)/cpp/external/filesystem-1.5.8/include
SHIM_CXXFLAGS := $(REF_CXXFLAGS) -I$(REPO)/include -I$(REPO)/integration

# Source list = the add_executable(katago ...) list of $(REF)/cpp/CMakeLists.txt:232-369 minus the
# backend TU (replaced by integration/katamxbackend.cpp).
REF_SRCS := \
  core/global.cpp core/base64.cpp core/bsearch.cpp core/commandloop.cpp core/config_parser.cpp core/datetime.cpp \
  core/elo.cpp core/fancymath.cpp core/fileutils.cpp core/hash.cpp core/logger.cpp core/mainargs.cpp core/makedir.cpp \
  core/md5.cpp core/multithread.cpp core/parallel.cpp core/rand.cpp core/rand_helpers.cpp core/sha2.cpp core/test.cpp \
  core/threadsafecounter.cpp core/threadsafequeue.cpp core/threadtest.cpp core/timer.cpp \
  game/board.cpp game/rules.cpp game/boardhistory.cpp game/graphhash.cpp \
  dataio/sgf.cpp dataio/numpywrite.cpp dataio/poswriter.cpp dataio/trainingwrite.cpp dataio/loadmodel.cpp \
  dataio/homedata.cpp dataio/files.cpp \
  neuralnet/nninputs.cpp neuralnet/sgfmetadata.cpp neuralnet/modelversion.cpp neuralnet/nneval.cpp neuralnet/desc.cpp \
  neuralnet/debugprint.cpp \
  book/book.cpp book/bookcssjs.cpp \
  search/timecontrols.cpp search/searchparams.cpp search/mutexpool.cpp search/search.cpp search/searchnode.cpp \
  search/searchresults.cpp search/searchhelpers.cpp search/searchexplorehelpers.cpp search/searchmirror.cpp \
  search/searchmultithreadhelpers.cpp search/searchnnhelpers.cpp search/searchtimehelpers.cpp \
  search/searchupdatehelpers.cpp search/asyncbot.cpp search/distributiontable.cpp search/localpattern.cpp \
  search/searchnodetable.cpp search/subtreevaluebiastable.cpp search/evalcache.cpp search/patternbonustable.cpp \
  search/analysisdata.cpp search/reportedsearchvalues.cpp \
  program/gtpconfig.cpp program/setup.cpp program/playutils.cpp program/playsettings.cpp program/play.cpp \
  program/selfplaymanager.cpp \
  tests/testboardarea.cpp tests/testboardbasic.cpp tests/testbook.cpp tests/testcommon.cpp tests/testconfig.cpp \
  tests/testmisc.cpp tests/testnnevalcanary.cpp tests/testpassalivesuicide.cpp tests/testrules.cpp tests/testscore.cpp \
  tests/testsgf.cpp tests/testsymmetries.cpp tests/testnninputs.cpp tests/testownership.cpp tests/testsearchcommon.cpp \
  tests/testsearchnonn.cpp tests/testsearch.cpp tests/testsearchv3.cpp tests/testsearchv8.cpp tests/testsearchv9.cpp \
  tests/testsearchmisc.cpp tests/testtime.cpp tests/testtrainingwrite.cpp tests/testnn.cpp tests/tinymodel.cpp \
  tests/tinymodeldata.cpp \
  distributed/client.cpp \
  command/commandline.cpp command/analysis.cpp command/benchmark.cpp command/contribute.cpp command/evalsgf.cpp \
  command/gatekeeper.cpp command/genbook.cpp command/gputest.cpp command/gtp.cpp command/match.cpp command/misc.cpp \
  command/runtests.cpp command/sandbox.cpp command/selfplay.cpp command/startposes.cpp command/tune.cpp \
  command/writetrainingdata.cpp \
  main.cpp

REF_OBJS := $(addprefix $(OBJ)/,$(REF_SRCS:.cpp=.o)) $(OBJ)/zipfile_zlib.o

$(OBJ)/%.o: $(REF)/cpp/%.cpp
	@mkdir -p $(dir $@)
	$(CXX) $(REF_CXXFLAGS) -c $< -o $@

# dataio/numpywrite.cpp keeps NumpyBuffer<T>; its NO_LIBZIP ZipFile (every member throws, numpywrite.cpp:240-264) is
# compiled under another name and integration/zipfile_zlib.cpp supplies ZipFile over zlib, so `selfplay` can write .npz.
$(OBJ)/dataio/numpywrite.o: $(REF)/cpp/dataio/numpywrite.cpp
	@mkdir -p $(dir $@)
	$(CXX) $(REF_CXXFLAGS) -DZipFile=ZipFileWithoutLibzip -c $< -o $@

$(OBJ)/zipfile_zlib.o: $(REPO)/integration/zipfile_zlib.cpp
	@mkdir -p $(dir $@)
	$(CXX) $(REF_CXXFLAGS) -Wall -Wextra -W -c $< -o $@

# ONE object file of the binding for every binary: what it is bound to is decided at link time - libkatamx.so (the MI355X backend;
# integration/Makefile) or the C ABI implemented on the CPU oracle (oracle/Makefile, test infrastructure).
$(OBJ)/katamxbackend.o: $(REPO)/integration/katamxbackend.cpp $(REPO)/integration/katamx_leaf.h $(REPO)/include/katamx.h
	@mkdir -p $(dir $@)
	$(CXX) $(SHIM_CXXFLAGS) -Wall -c $< -o $@

# This repo's NNEvaluator (no server threads: rows go from the callers' threads to the leaf batcher, results come back by ticket),
# featuriser (inputs version 7 as bit planes) and fibers (K leaves in flight per OS thread), linked INSTEAD OF the reference's
# neuralnet/nneval.cpp.
$(OBJ)/katamx_nneval.o: $(REPO)/integration/katamx_nneval.cpp $(REPO)/integration/katamx_nneval.h $(REPO)/integration/katamx_leaf.h $(REPO)/integration/katamx_fibers.h $(REPO)/integration/katamx_features.h
	@mkdir -p $(dir $@)
	$(CXX) $(SHIM_CXXFLAGS) -Wall -c $< -o $@

$(OBJ)/katamx_features.o: $(REPO)/integration/katamx_features.cpp $(REPO)/integration/katamx_features.h
	@mkdir -p $(dir $@)
	$(CXX) $(SHIM_CXXFLAGS) -Wall -Wextra -c $< -o $@

# The linker redirects search.cpp's calls of Search::performTaskWithThreads to the fibers' wrapper; the reference's object files are
# untouched.
$(OBJ)/katamx_fibers.o: $(REPO)/integration/katamx_fibers.cpp $(REPO)/integration/katamx_fibers.h $(REPO)/integration/katamx_leaf.h
	@mkdir -p $(dir $@)
	$(CXX) $(SHIM_CXXFLAGS) -Wall -c $< -o $@
FIBER_WRAP := -Wl,--wrap=_ZN6Search22performTaskWithThreadsEPSt8functionIFviEEi

OWN_NNEVAL_OUT  := $(OBJ)/neuralnet/nneval.o
OWN_NNEVAL_OBJS := $(OBJ)/katamx_nneval.o $(OBJ)/katamx_features.o $(OBJ)/katamx_fibers.o
