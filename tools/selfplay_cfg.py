#!/usr/bin/env python3
"""Writes a `selfplay` config with the SETTINGS of the reference's production self-play run on 18-block nets
(cpp/configs/training/selfplay8mainb18.cfg: visit counts :42-50,:100, rules :72-78, board sizes :76-78, komi :80-95, root noise and
temperatures :139-160, search constants :164-196) for ONE GPU - what BASELINE configs[2] and configs[4] name. Test / bench
infrastructure: the product reads no config.

    main_settings()                      the reference's values as a dict (numGameThreads 800 over 8 GPUs -> 100 per GPU, :65)
    write(path, **overrides)             the config file; overrides replace or add keys
    MIXED_9_13_19                        configs[4]: 9x9 / 13x13 / 19x19 games in a 19x19 data buffer
"""
import collections

# (key, value) in the order the reference's file groups them; device keys (cudaDeviceToUse...) are the caller's business
_MAIN = collections.OrderedDict([
    # logs
    ("logSearchInfo", "false"), ("logMoves", "false"), ("logGamesEvery", "20"), ("logToStdout", "true"),
    # data writing
    ("dataBoardLen", "19"), ("maxDataQueueSize", "2000"), ("maxRowsPerTrainFile", "20000"), ("firstFileRandMinProp", "0.15"),
    # forks, openings
    ("earlyForkGameProb", "0.04"), ("earlyForkGameExpectedMoveProp", "0.025"), ("forkGameProb", "0.01"), ("forkGameMinChoices", "3"),
    ("earlyForkGameMaxChoices", "12"), ("forkGameMaxChoices", "36"), ("sekiForkHackProb", "0.01"),
    ("initGamesWithPolicy", "true"), ("policyInitAreaProp", "0.08"), ("startPosesPolicyInitAreaProp", "0.0025"),
    ("compensateAfterPolicyInitProb", "0.5"), ("forkSidePositionProb", "0.020"),
    # visits
    ("cheapSearchProb", "0.75"), ("cheapSearchVisits", "350"), ("cheapSearchTargetWeight", "0.0"),
    ("reduceVisits", "true"), ("reduceVisitsThreshold", "0.9"), ("reduceVisitsThresholdLookback", "3"), ("reducedVisitsMin", "350"),
    ("reducedVisitsWeight", "0.1"),
    ("handicapAsymmetricPlayoutProb", "0.5"), ("normalAsymmetricPlayoutProb", "0.01"), ("maxAsymmetricRatio", "8.0"),
    ("minAsymmetricCompensateKomiProb", "0.4"),
    ("policySurpriseDataWeight", "0.5"), ("valueSurpriseDataWeight", "0.1"),
    ("estimateLeadProb", "0.50"), ("estimateLeadVisits", "10"), ("switchNetsMidGame", "true"), ("fancyKomiVarying", "true"),
    # match
    ("numGameThreads", "100"), ("maxMovesPerGame", "1600"),
    # rules
    ("koRules", "SIMPLE,POSITIONAL,SITUATIONAL"), ("scoringRules", "AREA,TERRITORY"), ("taxRules", "NONE,NONE,SEKI,SEKI,ALL"),
    ("multiStoneSuicideLegals", "false,true"), ("hasButtons", "false,false,true"),
    ("bSizes", "7,9,11,13,15,17,19,8,10,12,14,16,18"), ("bSizeRelProbs", "1,4,3,10,7,9,75,1,2,4,6,8,10"), ("allowRectangleProb", "0.10"),
    ("komiAuto", "True"), ("komiStdev", "1.0"), ("komiBigStdevProb", "0.05"), ("komiBigStdev", "12.0"), ("komiBiggerStdevProb", "0.005"),
    ("komiBiggerStdev", "45.0"),
    ("handicapProb", "0.10"), ("handicapCompensateKomiProb", "0.60"), ("forkCompensateKomiProb", "0.80"), ("sgfCompensateKomiProb", "0.85"),
    ("handicapKomiInterpZeroProb", "0.05"), ("sgfKomiInterpZeroProb", "0.15"),
    ("drawRandRadius", "0.5"), ("noResultStdev", "0.166666666"),
    # search limits
    ("maxVisits", "2000"), ("numSearchThreads", "1"),
    # evaluator
    ("nnMaxBatchSize", "192"), ("nnCacheSizePowerOfTwo", "24"), ("nnMutexPoolSizePowerOfTwo", "18"), ("numNNServerThreadsPerModel", "1"),
    ("nnRandomize", "true"),
    # root move selection and biases
    ("chosenMoveTemperatureEarly", "0.75"), ("chosenMoveTemperatureHalflife", "19"), ("chosenMoveTemperature", "0.15"),
    ("chosenMoveSubtract", "0"), ("chosenMovePrune", "1"),
    ("rootNoiseEnabled", "true"), ("rootDirichletNoiseTotalConcentration", "10.83"), ("rootDirichletNoiseWeight", "0.25"),
    ("rootDesiredPerChildVisitsCoeff", "2"), ("rootNumSymmetriesToSample", "4"),
    ("useLcbForSelection", "true"), ("lcbStdevs", "5.0"), ("minVisitPropForLCB", "0.15"),
    # internal search parameters
    ("winLossUtilityFactor", "1.0"), ("staticScoreUtilityFactor", "0.05"), ("dynamicScoreUtilityFactor", "0.30"),
    ("dynamicScoreCenterZeroWeight", "0.25"), ("dynamicScoreCenterScale", "0.50"), ("noResultUtilityForWhite", "0.0"),
    ("drawEquivalentWinsForWhite", "0.5"),
    ("rootEndingBonusPoints", "0.5"), ("rootPruneUselessMoves", "true"), ("rootPolicyTemperatureEarly", "1.5"), ("rootPolicyTemperature", "1.1"),
    ("cpuctExploration", "1.05"), ("cpuctExplorationLog", "0.28"), ("fpuReductionMax", "0.2"), ("rootFpuReductionMax", "0.0"),
    ("valueWeightExponent", "0.5"), ("subtreeValueBiasFactor", "0.30"), ("subtreeValueBiasWeightExponent", "0.8"),
    ("useNonBuggyLcb", "true"), ("useGraphSearch", "true"), ("fpuParentWeightByVisitedPolicy", "true"),
    ("fpuParentWeightByVisitedPolicyPow", "2.0"), ("numVirtualLossesPerThread", "1"),
])

# BASELINE configs[4]: "mixed 9x9 / 13x13 / 19x19 self-play with ownership + score heads, training-data npz write-out"
MIXED_9_13_19 = {"bSizes": "9,13,19", "bSizeRelProbs": "1,1,1", "allowRectangleProb": "0.0"}
# one board size only (BASELINE configs[2]: b18c384nbt on 19x19)
ONLY_19 = {"bSizes": "19", "bSizeRelProbs": "1", "allowRectangleProb": "0.0"}


def main_settings():
    return collections.OrderedDict(_MAIN)


def write(path, **overrides):
    s = main_settings()
    for k, v in overrides.items():
        s[k] = str(v)
    with open(path, "w") as f:
        f.write("# generated by tools/selfplay_cfg.py: the settings of the reference's selfplay8mainb18.cfg for one GPU\n")
        for k, v in s.items():
            f.write("%s = %s\n" % (k, v))
    return path


if __name__ == "__main__":
    import sys

    kv = dict(a.split("=", 1) for a in sys.argv[2:])
    print(write(sys.argv[1], **kv))
