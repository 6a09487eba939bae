// FACalcIwEqClasses_fast.cpp -- build helper, NOT part of the product or of the oracle.
//
// A drop-in definition of the reference's class FACalcIwEqClasses (declared in
// blingfirecompile.library/inc/FACalcIwEqClasses.h, defined in src/FACalcIwEqClasses.cpp) for ONE use: the call
// FADfaPack_triv::BuildEqs makes inside `fa_fsm2fsm_pack --remap-iws --iw-map=...` (src/FADfaPack_triv.cpp:1140-1180) when the
// BERT lexers are packed.  There the automaton is a FARSDfa_renum_iws view -- the ~1,000-symbol minimised DFA seen through the
// 1.13 M original input weights -- and the reference enumerates states x original weights (7e10 GetDest calls, hours).  Two
// original weights are equivalent iff the columns of their REDUCED symbols are equal, so this version compares the ~1,000 reduced
// columns (6e7 GetDest calls) and lifts the result; classes are numbered in order of first appearance over ascending original
// weight, exactly like FASplitSets::Classify numbers them.  Linked IN FRONT of libfsaCompile.a together with the reference's
// unchanged fa_fsm2fsm_pack.cpp (tools/modelbuild/README.md); verified byte-for-byte on wbd.bin and by rebuilding the checked-in
// bert_base_cased_tok.bin.
#include "blingfire-compile_src_pch.h"
#include "FAConfig.h"
#include "FACalcIwEqClasses.h"
#include "FAAllocatorA.h"
#include "FARSDfaA.h"
#include "FAMapA.h"
#define private public            // the view's two pointers (old DFA, new -> old weight map) are all that is read
#include "FARSDfa_renum_iws.h"
#undef private

#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

namespace BlingFire
{

FACalcIwEqClasses::FACalcIwEqClasses (FAAllocatorA * pAlloc) :
    m_pInDfa (NULL), m_pDfaSigma (NULL), m_pInNfa (NULL), m_pNfaSigma (NULL), m_IwBase (0), m_IwMax (0), m_NewIwBase (0),
    m_pIw2NewIw (NULL), m_MaxNewIw (-1), m_split_sets (pAlloc)
{
    m_e2iw.SetAllocator (pAlloc); m_e2iw.Create ();
    m_e2info.SetAllocator (pAlloc); m_e2info.Create ();
    m_set2id.SetAllocator (pAlloc); m_set2id.SetEncoder (&m_enc);
    m_iws.SetAllocator (pAlloc); m_iws.Create ();
    m_ows.SetAllocator (pAlloc); m_ows.Create ();
}
void FACalcIwEqClasses::SetIwBase (const int IwBase) { m_IwBase = IwBase; }
void FACalcIwEqClasses::SetIwMax (const int IwMax) { m_IwMax = IwMax; }
void FACalcIwEqClasses::SetNewIwBase (const int NewIwBase) { m_NewIwBase = NewIwBase; }
void FACalcIwEqClasses::SetRsDfa (const FARSDfaA * pInDfa) { m_pInDfa = pInDfa; }
void FACalcIwEqClasses::SetDfaSigma (const FAMealyDfaA * p) { m_pDfaSigma = p; }
void FACalcIwEqClasses::SetRsNfa (const FARSNfaA * p) { m_pInNfa = p; }
void FACalcIwEqClasses::SetNfaSigma (const FAMealyNfaA * p) { m_pNfaSigma = p; }
void FACalcIwEqClasses::SetIw2NewIw (FAMapA * p) { m_pIw2NewIw = p; }
const int FACalcIwEqClasses::GetMaxNewIw () const { return m_MaxNewIw; }
void FACalcIwEqClasses::Prepare () {}
void FACalcIwEqClasses::Clear () {}

static void unsupported (const char * why)
{
    fprintf (stderr, "FACalcIwEqClasses_fast: %s -- use the reference tool for this input\n", why);
    exit (3);
}

void FACalcIwEqClasses::Process ()
{
    if (!m_pInDfa || m_pInNfa || m_pDfaSigma || m_pNfaSigma || !m_pIw2NewIw) unsupported ("only a plain RS DFA is handled");
    const FARSDfa_renum_iws * pView = dynamic_cast < const FARSDfa_renum_iws * > (m_pInDfa);
    if (!pView || !pView->m_pDfa || !pView->m_pNewIw2Old) unsupported ("the automaton is not an --iw-map view");
    const FARSDfaA * pOld = pView->m_pDfa;
    const FAMapA * pNew2Old = pView->m_pNewIw2Old;

    // columns of the reduced symbols, compared exactly (std::map on the whole column)
    const int * pOldIws = NULL;
    const int OldCount = pOld->GetIWs (&pOldIws);
    const int MaxState = pOld->GetMaxState ();
    std::map < std::vector < int >, int > col2id;
    std::map < int, int > old2col;
    std::vector < int > col ((size_t) MaxState + 1);
    for (int k = 0; k < OldCount; ++k) {
        for (int s = 0; s <= MaxState; ++s) col [(size_t) s] = pOld->GetDest (s, pOldIws [k]);
        std::map < std::vector < int >, int >::iterator it = col2id.find (col);
        if (it == col2id.end ()) it = col2id.insert (std::make_pair (col, (int) col2id.size ())).first;
        old2col [pOldIws [k]] = it->second;
    }
    // a weight without a reduced symbol (or one the reduced DFA never uses) has the all -1 column
    for (int s = 0; s <= MaxState; ++s) col [(size_t) s] = -1;
    std::map < std::vector < int >, int >::iterator ite = col2id.find (col);
    if (ite == col2id.end ()) ite = col2id.insert (std::make_pair (col, (int) col2id.size ())).first;
    const int EmptyCol = ite->second;

    // lift to the view's alphabet; class ids in order of first appearance over ascending weight
    const int * pIws = NULL;
    const int Count = m_pInDfa->GetIWs (&pIws);
    std::vector < int > col2class (col2id.size (), -1);
    int Classes = 0;
    m_MaxNewIw = -1;
    for (int i = 0; i < Count; ++i) {
        const int Iw = pIws [i];
        if (m_IwBase > Iw || m_IwMax < Iw) { m_pIw2NewIw->Set (Iw, Iw); continue; }
        int c = EmptyCol;
        const int * pO = pNew2Old->Get (Iw);
        if (pO) { std::map < int, int >::const_iterator f = old2col.find (*pO); if (f != old2col.end ()) c = f->second; }
        if (col2class [(size_t) c] < 0) col2class [(size_t) c] = Classes++;
        const int NewIw = col2class [(size_t) c] + m_NewIwBase;
        m_pIw2NewIw->Set (Iw, NewIw);
        if (m_MaxNewIw < NewIw) m_MaxNewIw = NewIw;
    }
    fprintf (stderr, "FACalcIwEqClasses_fast: %d weights, %d reduced symbols, %d classes\n", Count, OldCount, Classes);
}

}
