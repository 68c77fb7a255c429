{-# LANGUAGE ForeignFunctionInterface #-}
{-# LANGUAGE RecordWildCards          #-}
-- |
-- Swim.Sim -- reference-side binding to libswimsim.so (include/swimsim.h).
--
-- SOURCE ONLY: this image has no GHC (SURVEY.md F4), so this module is delivered
-- uncompiled.  It imports the reference's own `Types` unchanged and offers the
-- tick simulator behind the same `Config` / `Gossip` / `Message` surface:
--
--   simulate :: SimConfig -> Source IO (Tick, Gossip)   -- events as `Broadcast (Suspect|Alive|Dead ..)`
--   stepN, memberView, firstDetection
--   encodeEnvelope / decodeEnvelope                      -- the wire codec of include/swimwire.h
--
-- Struct layouts come from Swim.Offsets, GENERATED from the C headers by scripts/gen_hs_offsets.py
-- (the stand-in for hsc2hs in an image without GHC): no offset in this file is typed by hand.
--
-- Cabal stanza to add to swim.cabal's library:
--   exposed-modules: ..., Swim.Sim, Swim.Offsets
--   extra-libraries: swimsim
--   include-dirs:    <repo>/include
--   build-depends:   ..., conduit, stm
module Swim.Sim
  ( SimConfig(..), Sim, Tick
  , defaultSimConfig, configureSim, stepN, scheduleFault
  , drainEvents, simulate, memberView, firstDetection, digest
  , encodeEnvelope, decodeEnvelope
  , stepShard, stepCluster
  , injectRumor, injectRumorCluster
  ) where

import           Control.Concurrent.MVar (MVar, newMVar, withMVar)
import           Control.Exception (SomeException, bracket, throwIO, try)
import           Control.Monad (forM, forM_, unless, when)
import           Data.IORef (newIORef, readIORef, writeIORef)
import           Control.Monad.IO.Class (liftIO)
import           Data.Conduit (Source, yield)
import           Data.Int (Int32, Int64)
import           Data.Word (Word16, Word32, Word64, Word8)
import qualified Data.ByteString as BS
import qualified Data.ByteString.Unsafe as BSU
import           Data.Char (chr, ord)
import           Foreign.C.String (CString, peekCString)
import           Foreign.C.Types (CInt (..), CSize (..))
import           Foreign.ForeignPtr (ForeignPtr, newForeignPtr, withForeignPtr)
import           Foreign.Marshal.Alloc (alloca, allocaBytes)
import           Foreign.Marshal.Array (allocaArray, peekArray, pokeArray)
import           Foreign.Marshal.Utils (fillBytes)
import           Foreign.Ptr (FunPtr, Ptr, freeHaskellFunPtr, nullPtr, plusPtr)
import qualified Foreign.Ptr
import           Foreign.Storable (peek, peekByteOff, poke, pokeByteOff)

import           Swim.Offsets
import           Types  -- the reference's src/Types.hs, unchanged

type Tick = Word64

-- | Simulator knobs around the reference 'Config' (numToGossip and gossipInterval are
-- honoured; bindHost / joinHosts / udpBufferSize are carried unused, as in the reference).
data SimConfig = SimConfig
  { simCfg            :: Config
  , simMembers        :: Word32
  , simSeed           :: Word64
  , simLossPpm        :: Word32
  , simSuspicionTicks :: Word32   -- 0 = 3 * ceil(log2 N)
  , simRetransmitMult :: Word32   -- 0 = 3
  , simMaxSubjects    :: Word32
  , simGcTicks        :: Word32   -- settling horizon (`removeDeadNodes`, src/Core.hs:65-67): 0 = off, maxBound = auto
  , simDevice         :: Int32
  , simTargetScheme   :: Word32   -- 0 = kRandomMembers (the reference), 1 = robust round-robin (src/Core.hs:232 FIXME)
  , simJoinPull       :: Word32   -- 1 = a member that comes up merges a join host's member map (`joinHosts`, src/Types.hs:47)
  , simPullTicks      :: Word32   -- T > 1 = every up member pulls a random up member's map once per T periods (the commented-out PushPullMsg, src/Types.hs:165,177); 0 = off
  , simViewCap        :: Word32   -- C > 0 = bounded member maps: at most C non-default entries per member, oldest evicted (`Map String Member`, src/Types.hs:55, with a capacity; include/swimsim.h); 0 = unbounded
  , simStrictReferenceRules :: Bool -- the literal suspectOrDeadNode' (src/Core.hs:142-187) under a canonical order instead of the commutative merge (D13; include/swimsim.h "Strict reference rules")
  , simPushPull       :: Bool     -- with simPullTicks: the periodic pull is a push-pull (the host merges the puller's map too: the push half of PushPullMsg, src/Types.hs:165,177)
  , simShardIndex     :: Word32   -- this handle's shard of a cluster of simNShards handles (one per GPU); 0 of 1 = unsharded
  , simNShards        :: Word32   -- 0 or 1 = one handle steps the whole population (stepN); > 1 = stepShard
  }

data SwimsimT
newtype Sim = Sim (MVar (ForeignPtr SwimsimT))   -- one handle = one logical thread of control

foreign import ccall unsafe "swimsim_default_config" c_default_config :: Ptr () -> IO CInt
foreign import ccall safe   "swimsim_create_msg"     c_create  :: Ptr () -> Ptr (Ptr SwimsimT) -> CString -> CSize -> IO CInt
foreign import ccall unsafe "&swimsim_destroy"       p_destroy :: FunPtr (Ptr SwimsimT -> IO ())
foreign import ccall unsafe "swimsim_last_error"     c_last_error :: Ptr SwimsimT -> IO CString
foreign import ccall safe   "swimsim_step"           c_step :: Ptr SwimsimT -> Word32 -> IO CInt   -- long-running => safe
foreign import ccall unsafe "swimsim_schedule_fault" c_fault :: Ptr SwimsimT -> Word64 -> Word32 -> Word8 -> IO CInt
foreign import ccall safe   "swimsim_drain_events"   c_drain :: Ptr SwimsimT -> Ptr () -> CSize -> Ptr CSize -> IO CInt
foreign import ccall safe   "swimsim_read_view"      c_view  :: Ptr SwimsimT -> Word32 -> Ptr () -> CSize -> Ptr CSize -> IO CInt
foreign import ccall safe   "swimsim_first_detect"   c_first :: Ptr SwimsimT -> Ptr Word64 -> CSize -> IO CInt
foreign import ccall safe   "swimsim_digest"         c_digest :: Ptr SwimsimT -> Ptr Word64 -> IO CInt
-- sharded clusters (one handle per GPU; include/swimsim.h "sharded clusters"): the caller moves the
-- records of the two exchange rounds between the handles (MPI / RCCL binding of the embedder's choice;
-- swim_amd/shard.py is the worked example over torch.distributed).
foreign import ccall unsafe "swimsim_shard_info"    c_shard_info    :: Ptr SwimsimT -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> IO CInt
foreign import ccall unsafe "swimsim_shard_buffers" c_shard_buffers :: Ptr SwimsimT -> Ptr (Ptr ()) -> Ptr (Ptr ()) -> IO CInt
-- settling on a sharded cluster: the kind-3 buffers of exchange round 3 (8-byte records, one list for every peer)
foreign import ccall unsafe "swimsim_shard_settle_buffers" c_shard_settle_buffers :: Ptr SwimsimT -> Ptr (Ptr ()) -> Ptr (Ptr ()) -> Ptr Word32 -> IO CInt
-- join_pull on a sharded cluster: the kind-4 buffers of exchange round 0 (16-byte records {joiner, subject, entry, -})
foreign import ccall unsafe "swimsim_shard_join_buffers" c_shard_join_buffers :: Ptr SwimsimT -> Ptr (Ptr ()) -> Ptr (Ptr ()) -> Ptr Word32 -> IO CInt
-- the replicas of a shard (queue masks / lines, queue bytes): the two tables all-gathered with exchange round 1
foreign import ccall unsafe "swimsim_shard_gather_buffers" c_shard_gather_buffers :: Ptr SwimsimT -> Ptr (Ptr ()) -> Ptr (Ptr ()) -> Ptr Word32 -> IO CInt
foreign import ccall safe   "swimsim_shard_phase1"  c_shard_phase1  :: Ptr SwimsimT -> Ptr Word32 -> IO CInt
foreign import ccall safe   "swimsim_shard_phase2"  c_shard_phase2  :: Ptr SwimsimT -> Ptr Word32 -> Ptr Word32 -> IO CInt
foreign import ccall safe   "swimsim_shard_phase3"  c_shard_phase3  :: Ptr SwimsimT -> Ptr Word32 -> Ptr Word32 -> IO CInt
-- the tick loop of a shard inside the library, the embedder lending it the all-to-all-v (swimsim_exchange_fn)
type ExchangeFn = Ptr () -> CInt -> Ptr Word32 -> Ptr Word32 -> IO CInt
foreign import ccall "wrapper" mkExchange :: ExchangeFn -> IO (FunPtr ExchangeFn)
foreign import ccall safe   "swimsim_shard_step"    c_shard_step    :: Ptr SwimsimT -> Word32 -> FunPtr ExchangeFn -> Ptr () -> IO CInt
-- a whole cluster of handles (dense or bounded) owned by this process, the exchange inside the library (include/swimsim.h)
foreign import ccall safe   "swimsim_cluster_step"  c_cluster_step  :: Ptr (Ptr SwimsimT) -> Word32 -> Word32 -> IO CInt
-- messages from outside the simulation (what `process` does with a Suspect / Alive / Dead off the socket, src/Core.hs:110-117)
foreign import ccall unsafe "swimsim_inject_rumor"       c_inject :: Ptr SwimsimT -> Word32 -> Word32 -> Word8 -> Word32 -> IO CInt
foreign import ccall unsafe "swimsim_note_outside_rumor" c_note   :: Ptr SwimsimT -> Word32 -> Word32 -> IO CInt

defaultSimConfig :: Config -> SimConfig
defaultSimConfig c = SimConfig c 128 1 0 0 0 0 0 0 0 0 0 0 False False 0 1

memberNameOf :: Word32 -> String
memberNameOf i = 'm' : show i

-- | `configure :: IO (Either Error Store)` (src/Util.hs:103-107) for the whole population.
configureSim :: SimConfig -> IO (Either Error Sim)
configureSim SimConfig{..} =
  allocaBytes swimsimConfigSize $ \p -> alloca $ \ph -> allocaBytes 512 $ \perr -> do
    _ <- c_default_config p
    pokeByteOff p swimsimConfig_num_to_gossip      (fromIntegral (numToGossip simCfg) :: Int32)
    pokeByteOff p swimsimConfig_gossip_interval_us (fromIntegral (gossipInterval simCfg) :: Int64)
    pokeByteOff p swimsimConfig_n_members          simMembers
    pokeByteOff p swimsimConfig_seed               simSeed
    pokeByteOff p swimsimConfig_loss_ppm           simLossPpm
    pokeByteOff p swimsimConfig_suspicion_ticks    simSuspicionTicks
    pokeByteOff p swimsimConfig_retransmit_mult    simRetransmitMult
    pokeByteOff p swimsimConfig_max_subjects       simMaxSubjects
    pokeByteOff p swimsimConfig_gc_ticks           simGcTicks
    pokeByteOff p swimsimConfig_device             simDevice
    pokeByteOff p swimsimConfig_target_scheme      simTargetScheme
    pokeByteOff p swimsimConfig_join_pull          simJoinPull
    pokeByteOff p swimsimConfig_pull_ticks         simPullTicks
    pokeByteOff p swimsimConfig_view_cap           simViewCap
    pokeByteOff p swimsimConfig_strict_reference_rules (if simStrictReferenceRules then 1 else 0 :: Word32)
    pokeByteOff p swimsimConfig_push_pull          (if simPushPull then 1 else 0 :: Word32)
    pokeByteOff p swimsimConfig_shard_index        simShardIndex
    pokeByteOff p swimsimConfig_n_shards           (max 1 simNShards)
    -- the failure text comes back through OUR buffer: the library's per-thread text could belong to another
    -- OS thread by the time a second `safe` call reads it
    rc <- c_create p ph perr 512
    if rc /= 0
      then Left <$> peekCString perr
      else do h <- peek ph
              fp <- newForeignPtr p_destroy h
              Right . Sim <$> newMVar fp

withSim :: Sim -> (Ptr SwimsimT -> IO a) -> IO a
withSim (Sim mv) f = withMVar mv (`withForeignPtr` f)

check :: Ptr SwimsimT -> CInt -> IO ()
check h rc = unless (rc == 0) $ c_last_error h >>= peekCString >>= ioError . userError

-- | nticks periods of `failureDetector` for every member (src/Core.hs:236-240).
stepN :: Sim -> Word32 -> IO ()
stepN s n = withSim s $ \h -> c_step h n >>= check h

scheduleFault :: Sim -> Tick -> Word32 -> Bool -> IO ()
scheduleFault s t m up = withSim s $ \h -> c_fault h t m (if up then 1 else 0) >>= check h

eventToGossip :: Ptr () -> IO (Tick, String, Gossip)
eventToGossip p = do
  t   <- peekByteOff p swimsimEvent_tick        :: IO Word64
  obs <- peekByteOff p swimsimEvent_observer    :: IO Word32
  sub <- peekByteOff p swimsimEvent_subject     :: IO Word32
  inc <- peekByteOff p swimsimEvent_incarnation :: IO Word32
  st  <- peekByteOff p swimsimEvent_state       :: IO Word8
  let name = memberNameOf sub
      msg = case st of
        1 -> Suspect { incarnation = fromIntegral inc, node = name }
        2 -> Dead { incarnation = fromIntegral inc, node = name, deadFrom = memberNameOf obs }
        _ -> Alive { incarnation = fromIntegral inc, node = name, addr = 0, port = 0 }
  return (t, memberNameOf obs, Broadcast msg)

-- | The `Broadcast` gossip the members enqueued since the last drain (src/Core.hs:119-121,254).
drainEvents :: Sim -> IO [(Tick, String, Gossip)]
drainEvents s = withSim s $ \h -> alloca $ \pn -> do
  poke pn 0
  rc0 <- c_drain h nullPtr 0 pn          -- 0: nothing to drain; ERR_BUFFER: pn = how many; anything else: a real error
  when (rc0 /= 0 && fromIntegral rc0 /= cSwimsimErrBuffer) $ check h rc0
  n <- peek pn
  if rc0 == 0 || n == 0 then return [] else
    allocaBytes (fromIntegral n * swimsimEventSize) $ \buf -> do
      c_drain h buf n pn >>= check h
      m <- peek pn
      forM [0 .. fromIntegral m - 1] $ \k -> eventToGossip (buf `plusPtr` (k * swimsimEventSize))

-- | Stream of membership events, one tick at a time, in the reference's `Source IO Gossip` style
-- (cf. `failureDetector :: Store -> Source IO Gossip`, src/Core.hs:233).
simulate :: Sim -> Word32 -> Source IO (Tick, Gossip)
simulate s ticks = go ticks
  where go 0 = return ()
        go k = do evs <- liftIO (stepN s 1 >> drainEvents s)
                  mapM_ (\(t, _, g) -> yield (t, g)) evs
                  go (k - 1)

-- | `members store` (src/Core.hs:76-77) for one observer: the non-default entries of its map.
memberView :: Sim -> Word32 -> IO [(String, Liveness, Int, Tick)]
memberView s obs = withSim s $ \h -> alloca $ \pn -> do
  poke pn 0
  rc0 <- c_view h obs nullPtr 0 pn
  when (rc0 /= 0 && fromIntegral rc0 /= cSwimsimErrBuffer) $ check h rc0
  n <- peek pn
  if rc0 == 0 || n == 0 then return [] else
    allocaBytes (fromIntegral n * swimsimViewEntrySize) $ \buf -> do
      c_view h obs buf n pn >>= check h
      forM [0 .. fromIntegral n - 1] $ \k -> do
        let p = buf `plusPtr` (k * swimsimViewEntrySize)
        sub <- peekByteOff p swimsimViewEntry_subject :: IO Word32
        inc <- peekByteOff p swimsimViewEntry_incarnation :: IO Word32
        since <- peekByteOff p swimsimViewEntry_since_tick :: IO Word32
        st <- peekByteOff p swimsimViewEntry_state :: IO Word8
        return (memberNameOf sub, toEnum (fromIntegral st), fromIntegral inc, fromIntegral since)

firstDetection :: Sim -> Word32 -> IO [Maybe Tick]
firstDetection s n = withSim s $ \h -> allocaArray (fromIntegral n) $ \buf -> do
  c_first h buf (fromIntegral n) >>= check h
  map (\v -> if v == maxBound then Nothing else Just v) <$> peekArray (fromIntegral n) buf

digest :: Sim -> IO Word64
digest s = withSim s $ \h -> alloca $ \p -> c_digest h p >>= check h >> peek p

-- ---------------------------------------------------------------------------------------------------
-- The wire codec (include/swimwire.h): `encode` / `decode` of `Envelope` (src/Types.hs:96-119) by the same
-- library -- for replaying simulated traffic into a live node, or parsing a live node's datagrams.
foreign import ccall unsafe "swimwire_encode"     c_wire_encode :: Ptr () -> CSize -> Ptr Word8 -> CSize -> Ptr CSize -> IO CInt
foreign import ccall unsafe "swimwire_decode"     c_wire_decode :: Ptr Word8 -> CSize -> Ptr () -> CSize -> Ptr CSize -> IO CInt
foreign import ccall unsafe "swimwire_last_error" c_wire_error  :: IO CString

-- Names travel as UTF-8 (as swim_amd/wire.py sends them); a name longer than the record's field, or an Ack payload longer
-- than its array, is REFUSED before anything is poked (`fits`): the C ABI promises to refuse over-long names, and a
-- poke past the 408-byte swimwire_msg_t would corrupt the next message or the allocaBytes buffer.
utf8 :: String -> [Word8]
utf8 = concatMap enc
  where enc ch | c < 0x80    = [fromIntegral c]
               | c < 0x800   = [0xC0 + fromIntegral (c `div` 64), 0x80 + fromIntegral (c `mod` 64)]
               | c < 0x10000 = [0xE0 + fromIntegral (c `div` 4096), 0x80 + fromIntegral ((c `div` 64) `mod` 64), 0x80 + fromIntegral (c `mod` 64)]
               | otherwise   = [0xF0 + fromIntegral (c `div` 262144), 0x80 + fromIntegral ((c `div` 4096) `mod` 64),
                                0x80 + fromIntegral ((c `div` 64) `mod` 64), 0x80 + fromIntegral (c `mod` 64)]
          where c = ord ch

unUtf8 :: [Word8] -> String
unUtf8 [] = []
unUtf8 (b : bs)
  | b < 0x80  = chr (fromIntegral b) : unUtf8 bs
  | b < 0xE0  = multi 1 (fromIntegral b - 0xC0)
  | b < 0xF0  = multi 2 (fromIntegral b - 0xE0)
  | otherwise = multi 3 (fromIntegral b - 0xF0)
  where multi k hi = let (cont, rest) = splitAt k bs
                     in chr (foldl (\acc x -> acc * 64 + (fromIntegral x - 0x80)) hi cont) : unUtf8 rest

nameFits :: String -> Bool
nameFits str = length (utf8 str) <= fromIntegral cSwimwireNameMax

-- | Nothing = the message fits a swimwire_msg_t; Just reason otherwise (checked by encodeEnvelope before any poke).
fits :: Message -> Maybe String
fits m = case m of
  Ping{..}         -> nm node
  IndirectPing{..} -> nm node
  Ack{..}          -> if length payload <= fromIntegral cSwimwirePayloadMax then Nothing else Just "Ack payload longer than SWIMWIRE_PAYLOAD_MAX"
  Suspect{..}      -> nm node
  Alive{..}        -> nm node
  Dead{..}         -> maybe (nm deadFrom) Just (nm node)
  where nm str = if nameFits str then Nothing else Just "node name longer than SWIMWIRE_NAME_MAX bytes of UTF-8"

pokeName :: Ptr () -> Int -> String -> IO ()
pokeName p off str = pokeArray (p `plusPtr` off) (utf8 str ++ [0 :: Word8])      -- `fits` has checked the length

peekName :: Ptr () -> Int -> IO String
peekName p off = unUtf8 . takeWhile (/= 0) <$> peekArray (fromIntegral cSwimwireNameMax + 1) (p `plusPtr` off :: Ptr Word8)

pokeMessage :: Ptr () -> Message -> IO ()
pokeMessage p m = do
  fillBytes p 0 swimwireMsgSize
  pokeByteOff p swimwireMsg_type (msgIndex m :: Word8)
  case m of
    Ping{..}         -> pokeByteOff p swimwireMsg_seq_no seqNo >> pokeName p swimwireMsg_node node
    IndirectPing{..} -> do pokeByteOff p swimwireMsg_seq_no seqNo; pokeByteOff p swimwireMsg_target target
                           pokeByteOff p swimwireMsg_port port; pokeName p swimwireMsg_node node
    Ack{..}          -> do pokeByteOff p swimwireMsg_seq_no seqNo
                           pokeByteOff p swimwireMsg_payload_len (fromIntegral (length payload) :: Word8)
                           pokeArray (p `plusPtr` swimwireMsg_payload) payload
    Suspect{..}      -> pokeByteOff p swimwireMsg_incarnation (fromIntegral incarnation :: Int64) >> pokeName p swimwireMsg_node node
    Alive{..}        -> do pokeByteOff p swimwireMsg_incarnation (fromIntegral incarnation :: Int64); pokeName p swimwireMsg_node node
                           pokeByteOff p swimwireMsg_addr addr; pokeByteOff p swimwireMsg_port port
    Dead{..}         -> do pokeByteOff p swimwireMsg_incarnation (fromIntegral incarnation :: Int64); pokeName p swimwireMsg_node node
                           pokeName p swimwireMsg_dead_from deadFrom

peekMessage :: Ptr () -> IO Message
peekMessage p = do
  ty  <- peekByteOff p swimwireMsg_type :: IO Word8
  sq  <- peekByteOff p swimwireMsg_seq_no
  inc <- fromIntegral <$> (peekByteOff p swimwireMsg_incarnation :: IO Int64)
  nd  <- peekName p swimwireMsg_node
  case ty of
    0 -> return (Ping sq nd)
    1 -> IndirectPing sq <$> peekByteOff p swimwireMsg_target <*> peekByteOff p swimwireMsg_port <*> pure nd
    2 -> do n <- peekByteOff p swimwireMsg_payload_len :: IO Word8
            Ack sq <$> peekArray (fromIntegral n) (p `plusPtr` swimwireMsg_payload)
    3 -> return (Suspect inc nd)
    4 -> Alive inc nd <$> peekByteOff p swimwireMsg_addr <*> peekByteOff p swimwireMsg_port
    _ -> Dead inc nd <$> peekName p swimwireMsg_dead_from

-- | `encode . Envelope` (src/Types.hs:96-103); Left = the codec's reason (more than 255 messages, > 65 535 bytes ..).
encodeEnvelope :: [Message] -> IO (Either String BS.ByteString)
encodeEnvelope msgs
  | (why : _) <- [w | Just w <- map fits msgs] = return (Left why)
  | otherwise =
  allocaBytes (max 1 (length msgs) * swimwireMsgSize) $ \arr -> allocaBytes (fromIntegral cSwimwireMaxDatagram) $ \buf -> alloca $ \pn -> do
    forM_' (zip [0 ..] msgs) $ \(k, m) -> pokeMessage (arr `plusPtr` (k * swimwireMsgSize)) m
    rc <- c_wire_encode arr (fromIntegral (length msgs)) buf (fromIntegral cSwimwireMaxDatagram) pn
    if rc /= 0 then Left <$> (c_wire_error >>= peekCString)
               else do n <- peek pn; Right <$> BS.packCStringLen (castPtr' buf, fromIntegral n)
  where forM_' xs f = mapM_ f xs
        castPtr' = Foreign.Ptr.castPtr

-- | `decode :: ByteString -> Either String Envelope` (src/Types.hs:105-119).
decodeEnvelope :: BS.ByteString -> IO (Either String [Message])
decodeEnvelope bytes = BSU.unsafeUseAsCStringLen bytes $ \(src, len) ->
  allocaBytes (fromIntegral cSwimwireMaxMsgs * swimwireMsgSize) $ \arr -> alloca $ \pn -> do
    rc <- c_wire_decode (Foreign.Ptr.castPtr src) (fromIntegral len) arr (fromIntegral cSwimwireMaxMsgs) pn
    if rc /= 0 then Left <$> (c_wire_error >>= peekCString)
               else do n <- peek pn
                       Right <$> forM [0 .. fromIntegral n - 1] (\k -> peekMessage (arr `plusPtr` (k * swimwireMsgSize)))

-- | `nticks` periods of this process's shard of a cluster of `nShards` handles (one per GPU / process), the
-- tick loop running inside the library (swimsim_shard_step): no Python in the loop.  `exchange round out`
-- is the embedder's all-to-all-v (MPI_Alltoallv, RCCL send/recv ..) over the buffers of
-- swimsim_shard_buffers: `out` holds 3 * nShards record counts (kind-major) to deliver, the result the
-- 3 * nShards counts that arrived.  With settling on (gcTicks) every tick ends with a round 3 over the buffers of
-- swimsim_shard_settle_buffers, and with joinPull a tick in which members come up starts with a round 0 over the
-- buffers of swimsim_shard_join_buffers: the counts of those rounds sit at indices [0 .. nShards).  Every shard of
-- the cluster must make the same call.  Round 1 is an all-gather: kind 0 (ONE segment for every peer: the tick's ring
-- dictionary and the queues that travel as lists; counts at [0 .. nShards)) and every shard's slice of the two tables of
-- swimsim_shard_gather_buffers (counts of the 8-byte kind at [nShards .. 2 nShards), of the 1-byte kind at
-- [2 nShards .. 3 nShards)); round 2 delivers the 8-byte records {dst, src} of kind 1 (counts at [nShards .. 2 nShards)).
-- The handle must have been configured as a shard (simShardIndex / simNShards).  A Haskell exception in `exchange` does
-- not cross the C frames: the callback reports failure (1), the library returns SWIMSIM_ERR_STATE and the exception is
-- re-thrown here; the FunPtr is freed on every path.
stepShard :: Sim -> Word32 -> Int -> (Int -> [Word32] -> IO [Word32]) -> IO ()
stepShard s nticks nShards exchange = withSim s $ \h -> do
  failure <- newIORef (Nothing :: Maybe SomeException)
  let body _ rnd pout pin = do
        r <- try $ do out <- peekArray (3 * nShards) pout
                      got <- exchange (fromIntegral rnd) out
                      pokeArray pin (take (3 * nShards) (got ++ repeat 0))
        case r of
          Right () -> return 0
          Left e   -> writeIORef failure (Just e) >> return 1
  rc <- bracket (mkExchange body) freeHaskellFunPtr $ \cb -> c_shard_step h nticks cb nullPtr
  readIORef failure >>= maybe (check h rc) throwIO

-- | All shards of a cluster in ONE process -- one GPU, or the GPUs of a node with peer access -- stepped together: the two
-- exchange rounds of a tick run on the handles' own streams (dense handles: the kernels read the peers' buffers in place;
-- bounded handles, simViewCap > 0: peer copies), nothing comes back to the host between the phases (`swimsim_cluster_step`;
-- DESIGN.md sections 7, 7b).  `sims` = shard 0, 1, ... of the cluster, in that order.  This is the host shape of the north
-- star: one Haskell process, eight GPUs, no exchange of its own to bring.
stepCluster :: [Sim] -> Word32 -> IO ()
stepCluster sims nticks = go sims []
  where
    go (s : rest) hs = withSim s $ \h -> go rest (h : hs)       -- every handle's lock is held for the call
    go [] hs = allocaArray (length hs) $ \arr -> do
      let ordered = reverse hs
      pokeArray arr ordered
      rc <- c_cluster_step arr (fromIntegral (length ordered)) nticks
      unless (rc == 0) $ firstError ordered rc
    firstError (h : rest) rc = do
      msg <- c_last_error h >>= peekCString
      if null msg && not (null rest) then firstError rest rc else ioError (userError (if null msg then "swimsim_cluster_step: status " ++ show rc else msg))
    firstError [] rc = ioError (userError ("swimsim_cluster_step: status " ++ show rc))

-- | A Suspect / Alive / Dead message about member @subject@ that reaches simulated member @observer@ from OUTSIDE the simulation
-- (`process`, src/Core.hs:110-117): delivered in the next tick stepped, ruled on like any rumour.  State as the reference's
-- constructor index (0 Alive, 1 Suspect, 2 Dead).
injectRumor :: Sim -> Word32 -> Word32 -> Word8 -> Word32 -> IO ()
injectRumor s observer subject st inc = withSim s $ \h -> c_inject h observer subject st inc >>= check h

-- | The same on a cluster (the shards in order, equal-sized contiguous id ranges): the owner of @observer@ takes the message, every
-- other shard is told of it (the subject's view row belongs to the whole cluster; include/swimsim.h).
injectRumorCluster :: [Sim] -> Word32 -> Word32 -> Word32 -> Word8 -> Word32 -> IO ()
injectRumorCluster sims nMembers observer subject st inc =
  forM_ (zip [0 ..] sims) $ \(k, s) -> withSim s $ \h ->
    if k == owner then c_inject h observer subject st inc >>= check h else c_note h observer subject >>= check h
  where owner = observer `div` (nMembers `div` fromIntegral (length sims)) :: Word32
