"""Host logic of the switching path that lives in blues/utils.py in the reference."""
import logging
import math
import sys

logger = logging.getLogger(__name__)


def _even_or_exit(n):
    """An NCMC protocol is symmetric about lambda = 0.5: an odd step count loses its last step, a count of 1 is refused."""
    if n % 2 == 0:
        return n
    even = n - 1
    text = "the symmetric NCMC protocol needs an even number of switching steps, got nstepsNC=%d" % n
    if even <= 0:
        logger.error(text)
        sys.exit(1)
    logger.warning("%s; using nstepsNC=%d" % (text, even))
    return even


def _next_even(x):
    k = int(x)
    return k if k % 2 == 0 else k + 1


def calculateNCMCSteps(nstepsNC=0, nprop=1, propLambda=0.3, **kwargs):
    """Contract of reference blues/utils.py:89-145 (checked against 1,512 cases computed by the reference's own function,
    tests/golden/reference_function_vectors.json).

    With `nprop` propagation passes per lambda increment inside the window 0.5 +/- propLambda and one outside, a protocol of L
    lambda increments takes  nprop * 2 floor(propLambda L) + 2 ceil((0.5 - propLambda) L)  integrator steps.  L is chosen as
    the even number at or above  nstepsNC / (2 (nprop propLambda + 0.5 - propLambda));  when the step count that follows
    is not the requested nstepsNC, the protocol is changed to L switching steps (as the reference does, with a warning).
    Returns {'nstepsNC', 'propSteps', 'moveStep', 'nprop', 'propLambda'}; the move fires half way."""
    nstepsNC = _even_or_exit(nstepsNC)
    inside, outside = propLambda, 0.5 - propLambda
    # (summed in the reference's order, nprop*propLambda + 0.5 - propLambda: the quotient is truncated, and 6 / 1.2000000000000002 is not 5)
    n_lambda = _next_even(nstepsNC / (2 * (nprop * propLambda + 0.5 - propLambda)))
    steps_inside = int(nprop * (2 * math.floor(inside * n_lambda)))
    steps_outside = int(2 * math.ceil(outside * n_lambda))
    propSteps = steps_inside + steps_outside
    if propSteps != nstepsNC:
        logger.warning("nstepsNC=%s cannot be met with propLambda=%s, nprop=%s: switching over %s lambda steps, %s propagation steps in all"
                       % (nstepsNC, propLambda, nprop, n_lambda, propSteps))
        nstepsNC = n_lambda
    return dict(nstepsNC=nstepsNC, propSteps=propSteps, moveStep=int(nstepsNC / 2), nprop=nprop, propLambda=propLambda)
