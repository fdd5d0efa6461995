"""Plain result records exchanged between an engine and the host drivers."""
import collections

# log-determinants of the covariances of set A, set B, their concatenation and,
# when asked for, of the count-weighted mean covariance used by GLR; `kl2` is the
# finished KL2 value (it needs pseudo-inverses, so it never leaves the engine as terms).
PairTerms = collections.namedtuple(
    'PairTerms', 'n1 n2 logdet1 logdet2 logdet_union logdet_glr kl2')

# events: ordered list of
#   ('cand', start, i, n1, n2, d, coarse)  candidate evaluation (only infinite ones
#                                           unless tracing was requested)
#   ('win', maxd_or_None)                   end of one coarse scan
#   ('det', start, maxi, d)                 accepted change point
GwTurnResult = collections.namedtuple('GwTurnResult', 'events final_start')

# merges: [(a, b, mind)] in the compacted indexing of the moment of the merge.
HiResult = collections.namedtuple('HiResult', 'merges max_dist min_dist')
